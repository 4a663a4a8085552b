"""Every kernel of the step must give bit-identical results whatever runs beside it on the other HIP streams.

Round 2 found one that did not: ``resize_fwd_kernel`` (csrc/conv.hip), compiled by hipcc -O3 into SLP-packed ``v_pk_*_f32``
arithmetic, returned different low halves of its float pairs in a few thousand elements WHENEVER the bf16 ViT-g teacher
(v_mfma_f32_32x32x16_bf16 waves) ran on the teacher stream at the same time - same input bits, no stray write anywhere
(guard bands around every allocation stayed intact, tests/diag/diag_guard_bands.py), fp32-MFMA co-runners harmless, and the
same source compiled with -fno-slp-vectorize immune (tests/diag/diag_student_vs_teacher.py, gpurun_out/r2i_*, r2k_*).  The
library is therefore built without SLP vectorisation (build.py), and this test keeps every launch wrapper under that
co-runner: reference bits with the chip otherwise idle, then repeated while the bf16 ViT-g forward is in flight."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_every_kernel_is_bit_reproducible_beside_the_bf16_teacher():
    from dinov2_distillation_amd import ops
    from dinov2_distillation_amd.models import DINOv2ViT
    dev = "cuda"
    torch.manual_seed(0)
    teacher = DINOv2ViT("dinov2_vitg14", seed=0, precision="bf16").cuda()
    xb = torch.randn(32, 3, 224, 224, device=dev)
    side = torch.cuda.Stream()
    with torch.no_grad():
        teacher(xb)
    torch.cuda.synchronize()

    def noise():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            teacher(xb)

    r = lambda *s: torch.randn(*s, device=dev)
    B = 32
    x4, x3, x1 = r(B, 7, 7, 1024), r(B, 14, 14, 512), r(B, 56, 56, 128)
    rows4, h4 = x4.view(-1, 1024), r(1568, 4096)
    g1k, b1k, g4k, b4k = r(1024), r(1024), r(4096), r(4096)
    y_ln, mu, rs = ops.layernorm_fwd(rows4, g1k, b1k, 1e-6)
    grn_y, grn_stats = ops.grn_fwd(h4, g4k, b4k, B, preact=True)
    bn_mean, bn_rstd = ops.bn_stats(rows4.clone())
    wt7, wt3 = r(49, 1024), r(9, 512)
    s, t = r(B, 256, 1536), r(B, 256, 1536)
    ms, mt = ops.colreduce(s, groups=B, scale=1 / 256), ops.colreduce(t, groups=B, scale=1 / 256)
    img = r(B, 3, 224, 224)
    idx = torch.randperm(1568, device=dev).int()
    mp_y, mp_arg = ops.maxpool3(x3)
    u12 = r(1568, 8192)
    qkv = r(B * 256, 3 * 1536)
    st = (256 * 3 * 1536, 3 * 1536)
    att = lambda: list(ops.attention_forward(qkv, qkv[:, 1536:], qkv[:, 3072:], 24, 64, 0.125, B, 256, 256, st, st, st))
    cases = {
        "resize_fwd 7->16": lambda: [ops.resize_fwd(x4, 16, 16)],
        "resize_fwd 14->16": lambda: [ops.resize_fwd(x3, 16, 16)],
        "resize_bwd 16->7": lambda: [ops.resize_bwd(dy16, 7, 7)],
        "dwconv7 fwd": lambda: [ops.dwconv7(x4, wt7, b1k)],
        "dwconv7 dgrad (flip, + add)": lambda: [ops.dwconv7(x4, wt7, None, add=x4, flip=True)],
        "dwconv7 wgrad": lambda: [ops.dwconv7_wgrad(x4, x4)],
        "dwconv7 fwd 56x56": lambda: [ops.dwconv7(x1, wt7_128, None)],
        "layernorm fwd": lambda: list(ops.layernorm_fwd(rows4, g1k, b1k, 1e-6)),
        "layernorm bwd": lambda: list(ops.layernorm_bwd(rows4, rows4, mu, rs, g1k)),
        "gelu fwd": lambda: [ops.gelu_fwd(h4)],
        "gelu bwd": lambda: [ops.gelu_bwd(h4, h4)],
        "relu bwd": lambda: [ops.relu_bwd(h4, h4)],
        "axpby": lambda: [ops.axpby(h4, h4, 0.5, 2.0)],
        "grn fwd (preact)": lambda: [ops.grn_fwd(h4, g4k, b4k, B, preact=True)[0]],
        "grn+gelu bwd": lambda: list(ops.grn_gelu_bwd(h4, None, h4, grn_stats, g4k, B)),
        "bn stats": lambda: list(ops.bn_stats(rows4.clone())),
        "bn+relu+pos fwd": lambda: [ops.bn_act_pos_fwd(rows4, bn_mean, bn_rstd, g1k, b1k, None, relu=True)],
        "bn bwd": lambda: list(ops.bn_bwd(rows4, rows4, bn_mean, bn_rstd, g1k, b1k, relu=True)),
        "colreduce sum / dot / gelusq": lambda: [ops.colreduce(h4), ops.colreduce(h4, ops.RED_DOT, y=h4), ops.colreduce(h4, ops.RED_GELUSQ, groups=B)],
        "transpose": lambda: [ops.transpose(h4)],
        "patchify nchw 14 / nhwc 2": lambda: [ops.patchify_nchw(img, 14), ops.patchify_nhwc(x3, 2)],
        "im2col 3x3 / col2im": lambda: [ops.im2col_nhwc(x3, 3, 1, 1), ops.col2im_nhwc(col3, B, 14, 14, 512, 3, 1, 1)],
        "implicit-GEMM conv fwd / wgrad (fp32), fwd (bf16)": lambda: [ops.conv_igemm_fwd(x3, w3i, 3, 1, 1), ops.conv_igemm_wgrad(x3, dz3i, 3, 1, 1),
                                                                  ops.conv_igemm16(x3_16, w3i_16, 3, 1, 1)],
        "dwconv3 fwd / dgrad / wgrad": lambda: [ops.dwconv3(x3, wt3, 2), ops.dwconv3(x3, wt3, 1, dgrad_shape=(14, 14)), ops.dwconv3_wgrad(x3, x3, 1)],
        "maxpool3 fwd / bwd": lambda: list(ops.maxpool3(x3)) + [ops.maxpool3_bwd(mp_y, mp_arg, 14, 14)],
        "avgpool2 fwd / bwd": lambda: [ops.avgpool2(x4), ops.avgpool2(ap_y, backward_hw=(7, 7))],
        "add_relu": lambda: [ops.add_relu(h4, h4)],
        "swiglu fwd / bwd": lambda: [ops.swiglu_fwd(u12), ops.swiglu_bwd(h4, u12)],
        "gather_rows / index_add_rows": lambda: [ops.gather_rows(rows4, idx), ops.index_add_rows(rows4, idx, 1568)],
        "dropout": lambda: [ops.dropout(h4, 0.3, 1234, 77)],
        "window_permute": lambda: [ops.window_permute(s[:, :, :384].contiguous(), 16, 16, 2, 2)],
        "kd_loss fwd / bwd (spatial)": lambda: [ops.kd_loss_fwd(s, t, 256, 0.08), ops.kd_loss_bwd(s, t, 256, 0.08)],
        "kd_loss fwd / bwd (frequency)": lambda: [ops.kd_loss_fwd(s, t, 256, 0.06, ms, mt), ops.kd_loss_bwd(s, t, 256, 0.06, None, ms, mt)],
        "sub_group_vec": lambda: [ops.sub_group_vec(s, ms, B)],
        "attention fwd hd64": att,
        "gemm f32 NT / NN / TN": lambda: [ops.gemm(rows4, w1), ops.gemm(h4, w1, b_layout=1), ops.gemm(h4, rows4, a_layout=1, b_layout=1)],
        "gemm f32 1568x1024x4096 + residual": lambda: [ops.gemm(h4, w2, bias=b1k, residual=rows4)],
        # ---- the kernels of the mixed-precision step (bf16 activations in HBM; ADVICE r02: they were not covered) ------------
        "gemm bf16 ring NT / NN / TN(+rowsum) / gelu / f32 resid": lambda: [
            ops.gemm_bf16(rows16, w1h, bias=b4h), ops.gemm_bf16(h16, w1h, b_layout=1),
            ops.gemm_bf16(h16, rows16, a_layout=1, b_layout=1, epi=ops.EPI_F32, rowsum=rs_buf), rs_buf,
            ops.gemm_bf16(rows16, w1h, epi=ops.EPI_BF16_GELU, bias=b4h),
            ops.gemm_bf16(h16, w2h, epi=ops.EPI_F32_RESID, bias=b1h, colscale=g1k, residual=rows4)],
        "gemm bf16 register-staged (ragged K)": lambda: [ops.gemm_bf16(rag16, wrag)],
        "cast bf16 / f32, round, add16, gelu16, relu_bwd16": lambda: [
            ops.cast_bf16(h4), ops.cast_f32(h16), ops.round_bf16_(h4.clone()), ops.add16(h16, h16), ops.gelu16(h16), ops.gelu16(h16, h16),
            ops.relu_bwd16(h16, h16)],
        "layernorm io fwd / bwd (bf16, cf16)": lambda: list(ops.layernorm_fwd_io(rows16, g1k, b1k, 1e-6, torch.bfloat16)) +
            list(ops.layernorm_fwd_io(rows16, g1k, b1k, 1e-6, torch.float32, cf16=True)) +
            list(ops.layernorm_bwd_io(rows16, rows16, mu, rs, g1k, torch.bfloat16)),
        "dwconv7 io fwd / dgrad / wgrad (bf16)": lambda: [
            ops.dwconv7_io(x4h, wt7, b1k), ops.dwconv7_io(x4h, wt7, None, add=x4h, flip=True),
            ops.dwconv7_wgrad_cw(x4h, x4h, round_out=True)],
        "colreduce16 / grn fwd16 / grn bwd16": lambda: [
            ops.colreduce16(h16, ops.RED_SUM, round_out=True), ops.grn_fwd16(h16, g4k, b4k, B)[0]] +
            list(ops.grn_gelu_bwd16(h16, h16, grn16_stats, g4k, B)),
        "attention16 fwd / bwd": lambda: list(ops.attention16_forward(q16, q16, q16, B, 16, 256, 256, 24, 24 ** -0.5, 5.0)) +
            list(ops.attention16_backward(q16, q16, q16, q16, a16[1], a16[2], B, 16, 256, 256, 24, 24 ** -0.5, 5.0)),
        "dct16 step / swiglu bf16": lambda: [ops.dct16_step(s16.view(B, 16, 16, 1536), dctm, 0), ops.swiglu_bf16(u16),
                                             ops.swiglu_bf16_bwd(h16, u16)],
        "prep_batch / colreduce_tr / grn_bwd_stats": lambda: [ops.colreduce_tr(part49, 49, round_out=True)],
    }
    b16 = lambda t: t.bfloat16()
    rows16, h16, x4h = b16(rows4), b16(h4), b16(x4)
    w1h, w2h, b4h, b1h = b16(r(4096, 1024) * 0.02), b16(r(1024, 4096) * 0.02), b16(r(4096)), b16(r(1024))
    rs_buf = torch.empty(4096, device=dev)
    rag16, wrag = b16(r(1000, 592)), b16(r(136, 592) * 0.05)
    grn16_stats = ops.grn_fwd16(h16, g4k, b4k, B)[1]
    q16 = b16(r(B * 256, 384))
    a16 = ops.attention16_forward(q16, q16, q16, B, 16, 256, 256, 24, 24 ** -0.5, 5.0)
    s16, u16, dctm = b16(s), b16(u12), r(16, 16)
    part49 = r(448, 49 * 1024)
    dy16, wt7_128, col3 = r(B, 16, 16, 1024), r(49, 128), r(B * 196, 9 * 512)
    w3i, dz3i = r(256, 9 * 512), r(B * 196, 256)
    x3_16, w3i_16 = x3.bfloat16(), w3i.bfloat16()
    ap_y = ops.avgpool2(x4)
    w1, w2 = r(4096, 1024) * 0.02, r(1024, 4096) * 0.02
    bad = []
    for name, f in cases.items():
        torch.cuda.synchronize()
        ref = [v.clone() for v in f()]
        torch.cuda.synchronize()
        for rep in range(3):
            noise()
            out = f()
            torch.cuda.synchronize()
            d = [i for i, (a, b) in enumerate(zip(out, ref)) if not torch.equal(a, b)]
            if d:
                n = int((out[d[0]] != ref[d[0]]).sum())
                bad.append(f"{name}: output {d[0]} differs in {n} elements beside the bf16 teacher (repetition {rep})")
                break
    assert not bad, "\n".join(bad)
