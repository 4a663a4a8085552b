"""The autocast policy on the 37 x 37 token grid of BASELINE configs[3] (518 px crops, 14 px patches): 1369 tokens are no multiple of
8, which the 16-bit attention core handles by zero-padding q / k / v and masking the padded keys (ops.attention16_*), and the
frequency loss's DCT Linears are 37 x 37 (dkd_dct_step).  Every node against torch.autocast itself on the oracle's op sequence
(CPU autocast, the CUDA fp32-op list forced), the yardstick being that oracle's distance from its own fp32 run."""
import pytest
import torch

pytestmark = pytest.mark.gpu

rel = lambda a, b: ((a.float().cpu().reshape(b.shape) - b.float().cpu()).norm() / b.float().norm()).item()


@pytest.mark.parametrize("Nq,Nk,shared", [(1369, 1369, False), (77, 1369, False), (1369, 1369, True)])
def test_attention16_ragged_token_counts_vs_torch_autocast(Nq, Nk, shared):
    from dinov2_distillation_amd import ops
    B, H, hd, c2 = 2, 4, 32, 5.0
    D = H * hd
    g = torch.Generator().manual_seed(Nq + Nk)
    q = torch.randn(1 if shared else B, Nq, D, generator=g).bfloat16()
    k, v = (torch.randn(B, Nk, D, generator=g).bfloat16() for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g).bfloat16()
    c1 = hd ** -0.5
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        qh = qr.expand(B, Nq, D).reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
        kh, vh = (t.reshape(B, Nk, H, hd).permute(0, 2, 1, 3) for t in (kr, vr))
        attn = (qh @ kh.transpose(-2, -1)) * c1 * c2
        with torch.autocast("cpu", enabled=False):
            attn = attn.float().softmax(dim=-1)
        ref = (attn @ vh).transpose(1, 2).reshape(B, Nq, D)
    ref.backward(do)
    qd = q.reshape(-1, D).cuda()
    kd, vd = (t.reshape(B * Nk, D).cuda() for t in (k, v))
    ops.ATTN16_FLASH = False   # this test is about the MATERIALISED path's padding (head dims 48 / 96 and A/B runs take it)
    try:
        O, S, P = ops.attention16_forward(qd, kd, vd, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
        assert O.shape == (B * Nq, D) and S.shape[-1] % 8 == 0 and S.shape[-2] % 8 == 0
        # the padded key columns carry no probability
        assert float(P[:, :, Nk:].float().abs().max() if P.shape[-1] > Nk else 0.0) == 0.0
        dq, dk, dv = ops.attention16_backward(do.reshape(B * Nq, D).cuda(), qd, kd, vd, S, P, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    finally:
        ops.ATTN16_FLASH = True
    qd64, kd64, vd64 = (t.double() for t in (q.expand(B, Nq, D), k, v))
    exact = (torch.softmax((qd64.reshape(B, Nq, H, hd).permute(0, 2, 1, 3) @ kd64.reshape(B, Nk, H, hd).permute(0, 2, 3, 1)) * c1 * c2, -1)
             @ vd64.reshape(B, Nk, H, hd).permute(0, 2, 1, 3)).transpose(1, 2).reshape(B, Nq, D)
    e, noise = rel(O, ref), rel(ref, exact)
    dq_ref = qr.grad
    if shared:   # the node returns the per-image query gradients; the repeat's adjoint (their sum) is the caller's
        dq = dq.float().reshape(B, Nq, D).sum(0)
    print(f"attention16 ({Nq} x {Nk}, shared {shared}): O ours vs torch autocast {e:.2e}; torch autocast vs exact {noise:.2e}; "
          f"dq {rel(dq, dq_ref):.2e} dk {rel(dk, kr.grad):.2e} dv {rel(dv, vr.grad):.2e}")
    assert e < 0.15 * noise and noise > 1e-3
    assert rel(dq, dq_ref) < 3e-3 and rel(dk, kr.grad) < 3e-3 and rel(dv, vr.grad) < 2e-3


def test_attention16_ragged_column_slices_of_a_packed_qkv_buffer():
    """the frozen-block path hands q / k / v (and dq / dk / dv) as column slices of one [rows, 3D] buffer: padded and cropped with
    that pitch, bit-equal to the dense call"""
    from dinov2_distillation_amd import ops
    B, H, hd, N = 2, 6, 64, 1369
    D = H * hd
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * N, 3 * D, generator=g).bfloat16().cuda()
    do = torch.randn(B * N, D, generator=g).bfloat16().cuda()
    q, k, v = (qkv[:, i * D:(i + 1) * D] for i in range(3))
    _packed_slices_check(ops, qkv, do, q, k, v, B, H, N, hd, D)            # flash kernels
    ops.ATTN16_FLASH = False
    try:
        _packed_slices_check(ops, qkv, do, q, k, v, B, H, N, hd, D)        # materialised path (padding copies with the slice pitch)
    finally:
        ops.ATTN16_FLASH = True


def _packed_slices_check(ops, qkv, do, q, k, v, B, H, N, hd, D):
    O1, S1, P1 = ops.attention16_forward(q, k, v, B, H, N, N, hd, 0.125, ld=3 * D)
    O2, S2, P2 = ops.attention16_forward(q.contiguous(), k.contiguous(), v.contiguous(), B, H, N, N, hd, 0.125)
    assert torch.equal(O1, O2) and torch.equal(P1, P2)
    dqkv = torch.full_like(qkv, float("nan"))
    ops.attention16_backward(do, q, k, v, S1, P1, B, H, N, N, hd, 0.125, ld=3 * D, out=tuple(dqkv[:, i * D:(i + 1) * D] for i in range(3)))
    d2 = ops.attention16_backward(do, q.contiguous(), k.contiguous(), v.contiguous(), S2, P2, B, H, N, N, hd, 0.125)
    assert torch.equal(dqkv, torch.cat(d2, dim=1))


@pytest.mark.parametrize("self_query", [True, False])
def test_projector_on_the_37_grid_under_the_policy_vs_torch_autocast(self_query):
    """losses/scalekd.py:228-245 at hw_dims = (37, 37) (configs[3]): forward and every gradient"""
    from oracle import scalekd_oracle as so
    from dinov2_distillation_amd.losses.scalekd import AttentionProjector
    D, P, B, cs, heads = 256, 37, 2, 96, 8
    g = torch.Generator().manual_seed(37 + self_query)
    m = AttentionProjector(cs, D, (P, P), D, self_query=self_query, softmax_scale=5.0, num_heads=heads)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if k.endswith("bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif "norm" in k or k.startswith("proj_student.1"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(B, cs, P, P, generator=g).bfloat16().float()
    query = None if self_query else torch.randn(B, P * P, D, generator=g)
    ct = torch.randn(B, P * P, D, generator=g)

    def oracle(policy):
        p = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        qr = None if query is None else query.clone().requires_grad_(True)
        if policy:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = so.attention_projector(p, "", xr, qr, (P, P), heads, 5.0, policy=True)
        else:
            out = so.attention_projector(p, "", xr, qr, (P, P), heads, 5.0)
        (out * ct).sum().backward()
        return p, xr, qr, out

    p16, x16, q16, o16 = oracle(True)
    p32, x32, q32, o32 = oracle(False)
    m.precision = "bf16_autocast"
    m.cuda().train()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2).requires_grad_(True)
    qg = None if query is None else query.cuda().requires_grad_(True)
    out = m(xg, query=qg)
    e, noise = rel(out, o16), rel(o16, o32)
    print(f"projector 37 x 37 (self_query {self_query}): tokens ours vs torch autocast {e:.2e}; torch autocast vs fp32 {noise:.2e}")
    assert out.dtype == torch.float32 and noise > 1e-3 and e < 0.3 * noise
    (out * ct.cuda()).sum().backward()
    bad = []
    pairs = [("x", xg.grad, x16.grad, x32.grad)] + ([] if query is None else [("query", qg.grad, q16.grad, q32.grad)])
    pairs += [(k, v.grad, p16[k].grad, p32[k].grad) for k, v in m.named_parameters()]
    for k, ours, a16, a32 in pairs:
        e, noise = rel(ours, a16), rel(a16, a32)
        print(f"  d{k}: ours vs torch autocast {e:.2e}; torch autocast vs fp32 {noise:.2e}")
        if e >= max(0.75 * noise, 1e-4):   # (same bound and reasoning as the 16 x 16 test, test_autocast_gpu.py)
            bad.append((k, e, noise))
    assert not bad, bad


def test_frequency_loss_on_the_37_grid_under_the_policy_vs_torch_autocast():
    """losses/scalekd.py:95-127 with DCT(resolution=37): four 37 x 37 16-bit Linears per map"""
    from oracle import scalekd_oracle as so
    from dinov2_distillation_amd.losses import ScaleKD
    B, P, D = 2, 37, 128
    g = torch.Generator().manual_seed(9)
    tok = torch.randn(B, P * P, D, generator=g)
    feat_t = torch.randn(B, D, P, P, generator=g) + 0.5
    m = ScaleKD(name="scalekd_res5", alpha=[0.08, 0.06], student_dims=96, teacher_dims=D, query_hw=[P, P], pos_hw=[P, P], pos_dims=D,
                self_query=True, softmax_scale=[5.0, 5.0], num_heads=8).set_precision("bf16_autocast")

    def oracle(policy):
        tr = tok.clone().requires_grad_(True)
        if policy:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                loss, sim = so.freq_loss(tr, feat_t, 0.06, policy=True)
        else:
            loss, sim = so.freq_loss(tr, feat_t, 0.06)
        loss.backward()
        return float(loss), float(sim), tr.grad

    l16, s16, g16 = oracle(True)
    l32, s32, g32 = oracle(False)
    tg = tok.cuda().requires_grad_(True)
    ft = feat_t.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2)
    loss, sim = m.get_freq_loss(tg, ft)
    loss.backward()
    e_l, n_l = abs(float(loss) - l16) / abs(l16), abs(l16 - l32) / abs(l32)
    e_g, n_g = rel(tg.grad, g16), rel(g16, g32)
    print(f"frequency loss 37 x 37 under the policy: loss ours vs torch autocast {e_l:.2e} (autocast vs fp32 {n_l:.2e}); "
          f"d(tokens) {e_g:.2e} (autocast vs fp32 {n_g:.2e}); similarity {float(sim):.3e} vs {s16:.3e}")
    assert e_l < max(0.2 * n_l, 2e-5) and e_g < 0.2 * n_g and n_g > 1e-3
    assert abs(float(sim) - s16) < 1e-4


def test_frozen_block_reuse_on_1369_tokens_under_the_policy_vs_torch_autocast():
    """train/distillation_module.py:176-177 at the 37 x 37 grid: functional16.ViTBlockFrozen16Fn on 1369 student tokens"""
    from oracle import vit_oracle as vo
    from dinov2_distillation_amd.models.dinov2 import Block
    gen = torch.Generator().manual_seed(5)
    D, H, B, N = 384, 6, 2, 1369
    blk = Block(D, H, gen)
    with torch.no_grad():
        for k, v in blk.state_dict().items():
            if k.endswith("gamma"):
                v.copy_(0.5 + torch.rand(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.copy_(0.05 * torch.randn(v.shape, generator=gen))
    p = {"blocks.0." + k: v.clone() for k, v in blk.state_dict().items()}
    blk.policy16 = True
    blk.cuda()
    tok, ct = torch.randn(B, N, D, generator=gen), torch.randn(B, N, D, generator=gen)

    def oracle(policy):
        xr = tok.clone().requires_grad_(True)
        if policy:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out = vo.vit_block_upstream_ops(p, "blocks.0.", xr, H, cuda_policy=True)
        else:
            out = vo.vit_block_upstream_ops(p, "blocks.0.", xr, H)
        (out * ct).sum().backward()
        return xr, out

    x16, o16 = oracle(True)
    x32, o32 = oracle(False)
    xg = tok.cuda().requires_grad_(True)
    out = blk(xg)
    e, noise = rel(out - xg.detach(), o16 - tok), rel(o16 - tok, o32 - tok)
    print(f"frozen block on 1369 tokens (added term): ours vs torch autocast {e:.2e}; torch autocast vs fp32 {noise:.2e}")
    assert noise > 1e-3 and e < 0.35 * noise
    (out * ct.cuda()).sum().backward()
    e, noise = rel(xg.grad - ct.cuda(), x16.grad - ct), rel(x16.grad - ct, x32.grad - ct)
    print(f"frozen block on 1369 tokens d(tokens) minus the residual path: ours vs torch autocast {e:.2e}; torch autocast vs fp32 {noise:.2e}")
    assert e < 0.5 * noise


@pytest.mark.parametrize("B,H,Nq,Nk,shared,c2,hd", [(3, 6, 256, 256, False, 1.0, 64), (2, 4, 77, 1369, False, 5.0, 64),
                                                   (2, 3, 1369, 1369, True, 5.0, 64), (2, 2, 64, 64, False, 5.0, 64),
                                                   (2, 16, 256, 256, True, 5.0, 24), (2, 24, 256, 256, False, 5.0, 16),
                                                   (2, 8, 1369, 1369, True, 5.0, 32), (3, 5, 100, 36, False, 1.0, 24),
                                                   (2, 16, 256, 256, True, 5.0, 48), (2, 16, 256, 256, True, 5.0, 96),
                                                   (2, 4, 1369, 1369, True, 5.0, 96), (3, 4, 196, 49, False, 1.0, 80),
                                                   (2, 3, 100, 77, False, 5.0, 72), (2, 5, 64, 64, False, 1.0, 40)])
def test_flash_policy_attention_equals_the_materialised_path(B, H, Nq, Nk, shared, c2, hd):
    """csrc/attention16_train.hip (hd 64: the frozen teacher blocks and the 16-head projector of configs[3]; hd 24 / 16: the projectors of
    configs[1]; 32: configs[3]'s second level; 96: the 16-head projector on configs[4]'s 1536 channels; 80 / 40: MiT) against the materialised path it replaces (batched bf16 GEMMs around
    dkd_attn16_softmax_*) and against torch.autocast: same rounding points, so the two differ by fp32 summation order only (a
    handful of 1-ulp bf16 flips), far inside the policy's distance from exact attention."""
    from dinov2_distillation_amd import ops
    D = H * hd
    g = torch.Generator().manual_seed(Nq + Nk + H)
    q = torch.randn(1 if shared else B, Nq, D, generator=g).bfloat16()
    k, v = (torch.randn(B, Nk, D, generator=g).bfloat16() for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g).bfloat16()
    c1 = hd ** -0.5
    qd = q.reshape(-1, D).cuda()
    kd, vd, dod = (t.reshape(-1, D).cuda() for t in (k, v, do))
    assert ops.ATTN16_FLASH
    O, m, linv = ops.attention16_forward(qd, kd, vd, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    assert m.dtype == torch.float32 and m.numel() == B * H * Nq     # row statistics, no [B*H, Nq, Nk] tensors
    dq, dk, dv = ops.attention16_backward(dod, qd, kd, vd, m, linv, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    ops.ATTN16_FLASH = False
    try:
        O2, S, P = ops.attention16_forward(qd, kd, vd, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
        dq2, dk2, dv2 = ops.attention16_backward(dod, qd, kd, vd, S, P, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    finally:
        ops.ATTN16_FLASH = True
    assert S.dtype == torch.bfloat16
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        qh = qr.expand(B, Nq, D).reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
        kh, vh = (t.reshape(B, Nk, H, hd).permute(0, 2, 1, 3) for t in (kr, vr))
        attn = (qh @ kh.transpose(-2, -1)) * c1 * c2
        with torch.autocast("cpu", enabled=False):
            attn = attn.float().softmax(dim=-1)
        ref = (attn @ vh).transpose(1, 2).reshape(B, Nq, D)
    ref.backward(do)
    exact = (torch.softmax((q.double().expand(B, Nq, D).reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
                            @ k.double().reshape(B, Nk, H, hd).permute(0, 2, 3, 1)) * c1 * c2, -1)
             @ v.double().reshape(B, Nk, H, hd).permute(0, 2, 1, 3)).transpose(1, 2).reshape(B, Nq, D)
    noise = rel(ref, exact)
    dqs = dq.float().reshape(B, Nq, D).sum(0) if shared else dq
    dq2s = dq2.float().reshape(B, Nq, D).sum(0) if shared else dq2
    print(f"flash policy attention ({Nq} x {Nk}, H {H}, hd {hd}, shared {shared}, c2 {c2}): vs materialised O {rel(O, O2.cpu()):.2e} dq {rel(dq, dq2.cpu()):.2e} "
          f"dk {rel(dk, dk2.cpu()):.2e} dv {rel(dv, dv2.cpu()):.2e}; vs torch autocast O {rel(O, ref):.2e} dq {rel(dqs, qr.grad):.2e} "
          f"dk {rel(dk, kr.grad):.2e} dv {rel(dv, vr.grad):.2e} (materialised: {rel(O2, ref):.2e} {rel(dq2s, qr.grad):.2e} "
          f"{rel(dk2, kr.grad):.2e} {rel(dv2, vr.grad):.2e}); torch autocast vs exact {noise:.2e}")
    assert torch.isfinite(O.float()).all() and torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all()
    assert rel(O, O2.cpu()) < 0.05 * noise and rel(O, ref) < 0.15 * noise and noise > 1e-3
    assert rel(dqs, qr.grad) < 3e-3 and rel(dk, kr.grad) < 3e-3 and rel(dv, vr.grad) < 2e-3
    assert rel(dq, dq2.cpu()) < 2e-3 and rel(dk, dk2.cpu()) < 2e-3 and rel(dv, dv2.cpu()) < 1e-3
    # deterministic
    O3, m3, l3 = ops.attention16_forward(qd, kd, vd, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    d3 = ops.attention16_backward(dod, qd, kd, vd, m3, l3, B, H, Nq, Nk, hd, c1, c2, q_shared=shared)
    assert torch.equal(O, O3) and all(torch.equal(x, y) for x, y in zip((dq, dk, dv), d3))

