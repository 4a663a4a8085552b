"""the reference's DEFAULT trainer setting is ``precision=16`` (train.py:263), i.e. torch autocast.  The teacher
forward has an autocast mode with bf16 activations in HBM (``DINOv2ViT(name, precision="bf16_autocast")``): every Linear on
bf16 operands with fp32 accumulation and a bf16 result, LayerNorm / softmax / residual stream in fp32.  Kernels against exact
emulations (one bf16 ulp where an fp32 accumulation order can flip a rounding), the teacher against the autocast oracle, which
tests/test_oracle_golden.py pins against torch.autocast itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

r16 = lambda t: t.bfloat16().float()


def ulp_close(out, ref, what, ulps=1.0, frac=1e-3):
    """bf16 results: equal up to `ulps` bf16 ulps of the reference (2^-8 relative), except for a fraction `frac` of elements
    where the fp32 accumulation order put the sum on the other side of a rounding boundary of an EARLIER rounding stage"""
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    tol = ulps * 2.0 ** -7 * ref.abs() + 1e-6 * ref.abs().max()
    bad = ((out - ref).abs() > tol).float().mean().item()
    worst = ((out - ref).abs() / (ref.abs().max() + 1e-30)).max().item()
    assert bad <= frac and worst < 2e-2, f"{what}: {bad:.2e} of the elements beyond {ulps} bf16 ulp, worst {worst:.2e} of the range"


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (8224, 384, 1536), (1000, 1152, 384), (513, 132, 592)])
def test_gemm_bf16_epilogues(M, N, K):
    from dinov2_distillation_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    w[:, 0] += torch.arange(N).bfloat16() * 0.01  # asymmetric: a transposed or row/column-swapped product cannot pass
    bias = torch.randn(N, generator=g).bfloat16()
    acc = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    out = ops.gemm_bf16(ad, wd, epi=ops.EPI_BF16, bias=bd)
    assert out.dtype == torch.bfloat16
    ulp_close(out, acc.float().bfloat16(), "bf16 out")
    if N % 128 == 0:
        og = ops.gemm_bf16(ad, wd, epi=ops.EPI_BF16_GELU, bias=bd)
        ulp_close(og, F.gelu(r16(acc.float())).bfloat16(), "bf16 gelu out", ulps=2.0, frac=5e-3)
    res = torch.randn(M, N, generator=g)
    ls = 0.5 + torch.rand(N, generator=g)
    o2 = ops.gemm_bf16(ad, wd, epi=ops.EPI_F32_RESID, bias=bd, colscale=ls.cuda(), residual=res.cuda())
    assert o2.dtype == torch.float32
    ref2 = res + ls * r16(acc.float())
    err = ((o2.cpu() - ref2).abs() > 2.0 ** -7 * (ls * r16(acc.float())).abs() + 1e-5).float().mean().item()
    assert err < 1e-3, err
    o3 = ops.gemm_bf16(ad, wd, epi=ops.EPI_F32, bias=bd)
    ulp_close(o3, r16(acc.float()), "fp32 out holding bf16 values")
    assert torch.equal(o3, o3.bfloat16().float())  # every stored value IS a bf16 value


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (8224, 384, 1536), (1000, 1152, 384), (520, 136, 592)])
def test_gemm_bf16_dgrad_and_wgrad_layouts(M, N, K):
    """the backward matmuls of a 16-bit Linear y[M,N] = x[M,K] W[N,K]^T with the operands AS THEY LIE: dx = dy W (W is the
    [K][N] operand of a reduction over N) and dW = dy^T x (both operands [rows][.], reduction over the rows; K-slabs when the
    small output cannot fill the chip).  Fragments come from transposing LDS reads: the operands are asymmetric on purpose."""
    from dinov2_distillation_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    w[:, 0] += torch.arange(N).bfloat16() * 0.01
    dy = torch.randn(M, N, generator=g).bfloat16()
    dy[:, 1] += (torch.arange(M) % 7).bfloat16() * 0.1
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    dx = ops.gemm_bf16(dyd, wd, b_layout=1)                                   # [M, K] = dy[M, N] . W[N, K]
    assert dx.dtype == torch.bfloat16 and dx.shape == (M, K)
    ulp_close(dx, (dy.double() @ w.double()).float().bfloat16(), "dgrad (NN)")
    dw = ops.gemm_bf16(dyd, xd, a_layout=1, b_layout=1, epi=ops.EPI_F32)     # [N, K] = dy^T x, fp32 holding the bf16 value
    assert dw.dtype == torch.float32 and dw.shape == (N, K)
    ref = (dy.double().t() @ x.double()).float()
    ulp_close(dw, r16(ref), "wgrad (TN), fp32 out holding bf16 values")
    assert torch.equal(dw, dw.bfloat16().float())
    dw16 = ops.gemm_bf16(dyd, xd, a_layout=1, b_layout=1)
    assert torch.equal(dw16.float(), dw) and torch.equal(ops.gemm_bf16(dyd, xd, a_layout=1, b_layout=1).float(), dw)  # reproducible
    yt = ops.gemm_bf16(xd.t().contiguous(), wd, a_layout=1)                  # A stored [K][M], B [N][K]
    ulp_close(yt, (x.double() @ w.double().t()).float().bfloat16(), "A km / B nk")


def test_gemm_bf16_wgrad_takes_k_slabs_on_step_shapes():
    from dinov2_distillation_amd import _lib, ops
    lib = _lib.load()
    assert lib.dkd_gemm_bf16_workspace_bytes(1536, 384, 12544) > 0 and lib.dkd_gemm_bf16_workspace_bytes(12544, 1536, 384) == 0
    g = torch.Generator().manual_seed(5)
    rows, n_out, k_in = 12544, 1536, 384
    dy, x = (torch.randn(rows, n_out, generator=g) * 0.1).bfloat16(), torch.randn(rows, k_in, generator=g).bfloat16()
    dw = ops.gemm_bf16(dy.cuda(), x.cuda(), a_layout=1, b_layout=1, epi=ops.EPI_F32)
    ulp_close(dw, r16((dy.double().t() @ x.double()).float()), "split wgrad")
    assert torch.equal(dw, ops.gemm_bf16(dy.cuda(), x.cuda(), a_layout=1, b_layout=1, epi=ops.EPI_F32))


def test_gemm_bf16_row_map_and_periodic_residual():
    """the patch-embed form: rows scattered past a CLS slot per image, position table added with period n"""
    from dinov2_distillation_amd import ops
    g = torch.Generator().manual_seed(3)
    B, n, K, D = 3, 16, 592, 128
    a = torch.randn(B * n, K, generator=g).bfloat16()
    w = (torch.randn(D, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(D, generator=g).bfloat16()
    pos = torch.randn(n, D, generator=g)
    out = torch.full((B * (n + 1), D), -7.0, device="cuda")
    ops.gemm_bf16(a.cuda(), w.cuda(), epi=ops.EPI_F32, bias=b.cuda(), residual=pos.cuda(), res_mod=n,
                  row_map=(n, n + 1, 1, B * (n + 1)), out=out)
    ref = r16((a.double() @ w.double().t() + b.double()).float()).view(B, n, D) + pos
    o = out.cpu().view(B, n + 1, D)
    assert (o[:, 0] == -7.0).all()  # the CLS rows are not touched
    assert ((o[:, 1:] - ref).abs().max() / ref.abs().max()).item() < 1e-2
    assert (((o[:, 1:] - ref).abs() > 2.0 ** -7 * ref.abs() + 1e-5).float().mean().item()) < 1e-3


def test_layernorm_bf16out_cast_swiglu():
    from dinov2_distillation_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(514, 1536, generator=g) * 3 + 1
    gm, bt = 1 + 0.1 * torch.randn(1536, generator=g), 0.1 * torch.randn(1536, generator=g)
    y = ops.layernorm_fwd_bf16out(x.cuda(), gm.cuda(), bt.cuda(), 1e-6)
    assert y.dtype == torch.bfloat16
    ulp_close(y, F.layer_norm(x.double(), (1536,), gm.double(), bt.double(), 1e-6).float().bfloat16(), "layernorm -> bf16")
    c = ops.cast_bf16(x.cuda())
    assert torch.equal(c.cpu(), x.bfloat16())
    xp = torch.randn(70, 588, generator=g)
    cp = ops.cast_bf16(xp.cuda(), 592).cpu()
    assert torch.equal(cp[:, :588], xp.bfloat16()) and (cp[:, 588:] == 0).all()
    u = torch.randn(300, 2 * 1024, generator=g).bfloat16()
    s = ops.swiglu_bf16(u.cuda())
    x1, x2 = u.float().chunk(2, -1)
    ulp_close(s, (r16(F.silu(x1)) * x2).bfloat16(), "swiglu bf16", ulps=1.0, frac=2e-3)


def test_attention_rounds_logits_and_probabilities_like_eager_autocast():
    from dinov2_distillation_amd import ops
    g = torch.Generator().manual_seed(4)
    B, N, H, hd = 2, 257, 6, 64
    D = H * hd
    qkv = r16(torch.randn(B * N, 3 * D, generator=g) * 1.5)
    st = (N * 3 * D, 3 * D)
    d = qkv.cuda()
    o_r, _ = ops.attention_forward(d, d[:, D:], d[:, 2 * D:], H, hd, hd ** -0.5, B, N, N, st, st, st, round_bf16=True)
    o_x, _ = ops.attention_forward(d, d[:, D:], d[:, 2 * D:], H, hd, hd ** -0.5, B, N, N, st, st, st)
    q, k, v = (qkv.view(B, N, 3, H, hd)[:, :, i].transpose(1, 2).double() for i in range(3))
    s = r16((q * hd ** -0.5 @ k.transpose(-1, -2)).float()).double()
    eager = (r16(torch.softmax(s, -1).float()).double() @ v).transpose(1, 2).reshape(B, N, D).float()
    exact = (torch.softmax(q * hd ** -0.5 @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B, N, D).float()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    e_r, e_x, gap = rel(o_r.cpu(), eager), rel(o_x.cpu(), exact), rel(eager, exact)
    print(f"rounded kernel vs eager-autocast emulation {e_r:.2e}; exact kernel vs exact {e_x:.2e}; emulation vs exact {gap:.2e}")
    assert e_x < 1e-5 and gap > 5e-4      # the rounding is a real, measurable change ...
    assert e_r < 0.35 * gap               # ... and the kernel makes THAT change (online softmax rounds p / running max, not p / final sum)


def _teacher(precision, seed_params=11):
    from dinov2_distillation_amd.models import DINOv2ViT
    t = DINOv2ViT("dinov2_vits14", seed=0, precision=precision)
    with torch.no_grad():
        gen = torch.Generator().manual_seed(seed_params)
        for k, v in t.model.state_dict().items():
            if k.endswith("gamma"):
                v.copy_(0.5 + torch.rand(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.copy_(0.02 * torch.randn(v.shape, generator=gen))
            elif "norm" in k and k.endswith("weight"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=gen))
    return t


def test_autocast_teacher_block_and_forward_vs_autocast_oracle():
    from oracle import vit_oracle as vo
    t = _teacher("bf16_autocast")
    p = {k: v.clone() for k, v in t.model.state_dict().items()}
    cfg = vo.VIT_CONFIGS["dinov2_vits14"]
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    t.cuda()
    # one block, on the term it adds to the residual stream
    tok = torch.randn(2, 257, 384, generator=torch.Generator().manual_seed(6))
    ref16 = vo.vit_block_autocast(p, "blocks.5.", tok, 6)
    ref32 = vo.vit_block(p, "blocks.5.", tok, 6)
    with torch.no_grad():
        out = t.model.blocks[5].forward_autocast(tok.cuda().reshape(-1, 384), 2, 257).view(2, 257, 384)
    e16, noise = rel(out.cpu() - tok, ref16 - tok), rel(ref16 - tok, ref32 - tok)
    print(f"autocast block (added term): ours vs autocast oracle {e16:.2e}; autocast oracle vs fp32 oracle {noise:.2e}")
    # (half of what is left is the flash kernel's online softmax: it rounds exp(s - running max) where eager attention rounds
    # exp(s - max) / sum - the same relative rounding applied to a differently scaled number, i.e. uncorrelated bf16 noise; the
    # attention test above measures that part alone at 0.25 of the rounding's own effect)
    assert noise > 2e-3 and e16 < 0.6 * noise
    # the whole teacher
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    f16 = vo.teacher_feature_map_autocast(p, x, cfg)
    f32 = vo.teacher_feature_map(p, x, cfg)
    with torch.no_grad():
        ours = t(x.cuda())["feature_map"]
    assert ours.dtype == torch.float32 and ours.shape == (2, 384, 16, 16) and ours.permute(0, 2, 3, 1).is_contiguous()
    e16, e32, noise = rel(ours, f16), rel(ours, f32), rel(f16, f32)
    print(f"autocast teacher: ours vs autocast oracle {e16:.2e}, vs fp32 oracle {e32:.2e}; autocast oracle vs fp32 oracle {noise:.2e}")
    assert e16 < 0.85 * noise and 0.5 * noise < e32 < 1.5 * noise
    # the mode is a property of the frozen teacher's own forward; block re-use on student tokens keeps its gradient path
    tg = tok.cuda().requires_grad_(True)
    t.model.blocks[9](tg).sum().backward()
    assert tg.grad is not None and torch.isfinite(tg.grad).all()


def test_step_with_autocast_teacher_stays_close_to_the_fp32_step():
    from dinov2_distillation_amd.trainer import DistillationStepper
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(77)).cuda()
    l32 = float(DistillationStepper(device="cuda:0").step(x).detach())
    st = DistillationStepper(device="cuda:0", teacher_precision="bf16_autocast")
    l16 = float(st.step(x).detach())
    l16b = float(st.step(x).detach())
    print(f"step loss: fp32 teacher {l32:.5f}, autocast-bf16 teacher {l16:.5f}")
    assert abs(l16 - l32) <= 2e-2 * abs(l32) and l16 != l32 and l16b == l16b
