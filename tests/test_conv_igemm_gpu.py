"""Implicit-GEMM k x k convolution (dkd_conv_igemm_f32: im2col addresses generated in the LDS-DMA operand loader) against
(1) a plain fp64 ``F.conv2d`` on the CPU - the op models/backbones/resnet.py:258-289 / stdc.py:8-24 call - forward, input gradient
and weight gradient, and (2) the materialised path it replaces (im2col + GEMM + col2im) on the same inputs.  Tolerance: fp32
summation-order noise only (2e-5 of the tensor's magnitude); edge cases: ragged pixel counts (clamped rows), an output width that
is no multiple of the tile, padding on every side, strided 1 x 1, 5 x 5, pixel counts the weight-gradient kernel refuses."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from dinov2_distillation_amd import ops as o
    return o


def close(a, b, what, rtol=2e-5):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    scale = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= rtol * scale, f"{what}: max|diff| {err:.3e} > {rtol:g} * {scale:.3e}"


CASES = [  # B, H, W, C, O, k, stride, pad
    (2, 14, 14, 64, 64, 3, 1, 1),
    (4, 28, 28, 32, 96, 3, 2, 1),      # 1.5 column tiles, stride 2
    (2, 9, 9, 64, 128, 3, 1, 1),       # 162 pixels: ragged row tile; no whole 32-pixel k-tiles for the weight gradient
    (3, 16, 16, 128, 64, 1, 2, 0),     # strided 1 x 1 (ResNet downsample shortcut)
    (2, 8, 8, 32, 32, 5, 1, 2),        # 5 x 5
    (8, 14, 14, 256, 256, 3, 1, 1),    # K = 2304: many k-tiles, taps change every 8 tiles
    (2, 7, 7, 512, 512, 3, 1, 1),      # 98 pixels, K = 4608: the split-K forward (few tiles)
]


def w2_of(w, k):
    O, C = w.shape[:2]
    return w.permute(0, 2, 3, 1).reshape(O, k * k * C).contiguous()


@pytest.mark.parametrize("B,H,W,C,O,k,s,p", [(64, 56, 56, 64, 64, 3, 1, 1),     # ResNet layer1's 3 x 3 at bs 64: 1568 tiles
                                             (64, 57, 57, 64, 128, 3, 2, 1),    # stride 2, odd image, ragged last row tile
                                             (40, 40, 40, 32, 96, 5, 1, 2)])    # 5 x 5, 32 channels (a tap per k-tile), ragged N tile
def test_igemm_x3_forward_and_stride1_dgrad(ops, monkeypatch, B, H, W, C, O, k, s, p):
    """the split-precision form of the implicit-GEMM convolution (csrc/gemm_x3.hip, gathered A operand; taken for chip-filling
    geometries in the default GEMM mode): forward with bias and - stride 1 - the input gradient through the flipped kernel, against
    fp64 convolutions at fp32 round-off level"""
    monkeypatch.setattr(ops, "GEMM_PRECISION", ops.PREC_BF16X3)
    from dinov2_distillation_amd import _lib
    assert _lib.load().dkd_conv_igemm_x3_supported(B, H, W, C, O, k, s, p) == 1
    g = torch.Generator().manual_seed(B + H + C + O + k)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(O, C, k, k, generator=g) / (k * C ** 0.5)
    bias = torch.randn(O, generator=g)
    y_ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=s, padding=p)
    Ho, Wo = y_ref.shape[2:]
    z = ops.conv_igemm_fwd(x.cuda(), w2_of(w, k).cuda(), k, s, p, bias=bias.cuda())
    close(z.view(B, Ho, Wo, O), y_ref.permute(0, 2, 3, 1), "x3 forward + bias", rtol=5e-6)
    if s == 1:
        dy = torch.randn(B, Ho, Wo, O, generator=g)
        dx_ref = torch.nn.grad.conv2d_input((B, C, H, W), w.double(), dy.double().permute(0, 3, 1, 2), stride=1, padding=p)
        dx = ops.conv_igemm_dgrad(dy.cuda(), w.cuda(), (B, H, W, C), k, p)
        close(dx, dx_ref.permute(0, 2, 3, 1), "x3 stride-1 input gradient", rtol=5e-6)


@pytest.mark.parametrize("B,H,W,C,O,k,s,p", CASES)
def test_igemm_forward_dgrad_wgrad(ops, B, H, W, C, O, k, s, p):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C + O + k)
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(O, C, k, k, generator=g) / (k * C ** 0.5)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    Ho, Wo = y_ref.shape[2:]
    assert ops.conv_igemm_ok(B, H, W, C, O, k, s, p)
    xg, w2 = x.cuda(), w2_of(w, k).cuda()
    z = ops.conv_igemm_fwd(xg, w2, k, s, p)
    close(z.view(B, Ho, Wo, O), y_ref.permute(0, 2, 3, 1), "forward")
    # the path it replaces, same inputs
    rows = ops.im2col_nhwc(xg, k, s, p)
    close(z, ops.gemm(rows, w2), "forward vs im2col + GEMM", rtol=2e-6)
    # with a bias in the epilogue
    bias = torch.randn(O, generator=g).cuda()
    close(ops.conv_igemm_fwd(xg, w2, k, s, p, bias=bias), z + bias, "bias epilogue", rtol=1e-6)
    dz = dy.permute(0, 2, 3, 1).reshape(-1, O).float().contiguous().cuda()
    if ops.conv_igemm_ok(B, H, W, C, O, k, s, p, 1):
        dw2 = ops.conv_igemm_wgrad(xg, dz, k, s, p)
        close(dw2.view(O, k, k, C).permute(0, 3, 1, 2), wd.grad, "weight gradient")
        close(dw2, ops.gemm(dz, rows, a_layout=1, b_layout=1), "weight gradient vs the materialised path", rtol=2e-6)
    else:
        assert (B * Ho * Wo) % 32 != 0
        with pytest.raises(Exception, match="UNSUPPORTED|status -3"):
            ops.conv_igemm_wgrad(xg, dz, k, s, p)
    if s == 1:
        wf = w2.view(O, k * k, C).flip(1).permute(2, 1, 0).reshape(C, k * k * O).contiguous()
        dx = ops.conv_igemm_fwd(dz.view(B, Ho, Wo, O), wf, k, 1, k - 1 - p).view(B, H, W, C)
        close(dx, xd.grad.permute(0, 2, 3, 1), "input gradient (forward of the flipped kernel)")
        close(dx, ops.col2im_nhwc(ops.dgrad(dz, w2), B, H, W, C, k, s, p), "input gradient vs GEMM + col2im", rtol=2e-6)


def test_igemm_refuses_what_it_cannot_tile(ops):
    assert not ops.conv_igemm_ok(2, 8, 8, 24, 64, 3, 1, 1)        # a 32-deep k-tile would straddle two taps
    assert not ops.conv_igemm_ok(2, 8, 8, 3, 64, 7, 2, 3)         # the RGB stem stays on im2col_nchw
    assert not ops.conv_igemm_ok(2, 9, 9, 64, 64, 3, 1, 1, 1)     # 162 pixels: weight gradient needs whole 32-pixel tiles
    x = torch.randn(2, 8, 8, 24, device="cuda")
    with pytest.raises(Exception, match="UNSUPPORTED|status -3"):
        ops.conv_igemm_fwd(x, torch.randn(64, 9 * 24, device="cuda"), 3, 1, 1)


@pytest.mark.parametrize("relu", [1, 3])
def test_conv_bn_act_node_implicit_equals_materialised(ops, relu):
    """the autograd node of the conv students (functional.ConvBNActFn) with and without the implicit kernel: same outputs, same
    gradients (fp32 summation-order noise), at a ResNet layer-3 shape (bs 8) and a stride-2 STDC shape"""
    from dinov2_distillation_amd import functional as Fn
    for (B, H, C, O, k, s) in [(8, 14, 256, 256, 3, 1), (4, 28, 64, 128, 3, 2)]:
        g = torch.Generator().manual_seed(7 * H + O + relu)
        x0 = torch.randn(B, H, H, C, generator=g).cuda()
        w0 = (torch.randn(O, C, k, k, generator=g) / (k * C ** 0.5)).cuda()
        gam, bet = (1 + 0.1 * torch.randn(O, generator=g)).cuda(), (0.1 * torch.randn(O, generator=g)).cuda()
        outs = []
        for flag in (True, False):
            Fn.CONV_IGEMM = flag
            try:
                x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
                gw, gb = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
                state = (torch.zeros(O).cuda(), torch.ones(O).cuda(), 0.1, 1e-5)
                y = Fn.ConvBNActFn.apply(x, w, gw, gb, state, k, s, relu, True, False)
                dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda()
                y.backward(dy)
                torch.cuda.synchronize()
                outs.append((y.detach(), x.grad, w.grad, gw.grad, gb.grad))
            finally:
                Fn.CONV_IGEMM = True
        for a, b, what in zip(outs[0], outs[1], ("y", "dx", "dw", "dgamma", "dbeta")):
            close(a, b, f"{what} [{B},{H},{C}->{O},s{s}]", rtol=5e-5)


# ---- the same kernel idea on the bf16 ring GEMM (the policy's convs) ---------------------------------------------------------------
CASES16 = [  # B, H, W, C, O, k, stride, pad
    (2, 14, 14, 64, 64, 3, 1, 1),
    (4, 28, 28, 64, 96, 3, 2, 1),      # stride 2, 0.75 column tiles
    (2, 9, 9, 128, 128, 3, 1, 1),      # ragged row tile (162 pixels)
    (3, 16, 16, 128, 64, 1, 2, 0),     # strided 1 x 1
    (8, 14, 14, 256, 256, 3, 1, 1),
    (2, 7, 7, 512, 512, 3, 1, 1),      # K = 4608 on 4 tiles: K-slabs + fold
    (2, 28, 28, 32, 64, 3, 1, 1),      # C = 32 (STDC, DarkNet): the 32-deep ring configuration, K = 288
    (4, 14, 14, 96, 96, 3, 2, 1),      # C = 96: 32-deep tiles, three per tap
]


def ulps16(a, b):
    """largest difference in units of the bf16 spacing at the reference's magnitude"""
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    # (results that cancel to near zero carry the fp32 accumulation noise of their O(1) partial sums: many spacings of a tiny number -
    # measure those against the spacing at 5 % of the tensor's mean magnitude)
    spacing = torch.clamp(b.abs(), min=0.05 * b.abs().mean().item()).log2().floor().exp2() * 2.0 ** -7
    return ((a - b).abs() / spacing).max().item()


@pytest.mark.parametrize("B,H,W,C,O,k,s,p", CASES16)
def test_igemm16_forward_and_dgrad(ops, B, H, W, C, O, k, s, p):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C + O + k + 1)
    x16 = torch.randn(B, H, W, C, generator=g).bfloat16()
    w16 = (torch.randn(O, C, k, k, generator=g) / (k * C ** 0.5)).bfloat16()
    y_ref = F.conv2d(x16.double().permute(0, 3, 1, 2), w16.double(), stride=s, padding=p)      # exact products, fp64 sums
    Ho, Wo = y_ref.shape[2:]
    assert ops.conv_igemm16_ok(B, H, W, C, O, k, s, p)
    xg = x16.cuda()
    w2 = w16.permute(0, 2, 3, 1).reshape(O, k * k * C).contiguous().cuda()
    z = ops.conv_igemm16(xg, w2, k, s, p)
    assert z.dtype == torch.bfloat16
    assert ulps16(z.view(B, Ho, Wo, O), y_ref.permute(0, 2, 3, 1)) <= 1.0 + 1e-6, "forward: more than one rounding away from fp64"
    rows16 = ops.im2col_nhwc(xg.view(torch.float32).reshape(B, H, W, C // 2), k, s, p).view(torch.bfloat16)
    z_mat = ops.gemm_bf16(rows16, w2)
    assert ulps16(z, z_mat) <= 1.0 + 1e-6 and (z != z_mat).float().mean().item() < 1e-3, "forward vs im2col + GEMM"
    z32 = ops.conv_igemm16(xg, w2, k, s, p, epi=ops.EPI_F32)
    assert z32.dtype == torch.float32 and torch.equal(z32, z.float()), "EPI_F32 = the same rounded values held in fp32"
    if s == 1:
        dy16 = torch.randn(B, Ho, Wo, O, generator=g).bfloat16()
        dx_ref = torch.nn.grad.conv2d_input((B, C, H, W), w16.double(), dy16.double().permute(0, 3, 1, 2), stride=1, padding=p)
        wf = w2.view(O, k * k, C).flip(1).permute(2, 1, 0).reshape(C, k * k * O).contiguous()
        dx = ops.conv_igemm16(dy16.cuda(), wf, k, 1, k - 1 - p)
        assert ulps16(dx.view(B, H, W, C), dx_ref.permute(0, 2, 3, 1)) <= 1.0 + 1e-6, "input gradient: more than ONE rounding"
        drows = ops.gemm_bf16(dy16.cuda().view(-1, O), w2, b_layout=1, epi=ops.EPI_F32_RAW)
        dx_mat = ops.cast_bf16(ops.col2im_nhwc(drows, B, H, W, C, k, s, p))
        assert ulps16(dx.view(-1, C), dx_mat) <= 1.0 + 1e-6 and (dx.view(-1, C) != dx_mat).float().mean().item() < 5e-3


def test_igemm16_refuses_what_it_cannot_tile(ops):
    assert not ops.conv_igemm16_ok(2, 8, 8, 48, 64, 3, 1, 1)      # a 32-deep k-tile would straddle two taps
    assert not ops.conv_igemm16_ok(2, 8, 8, 64, 60, 3, 1, 1)      # 16-byte output pieces
    with pytest.raises(Exception, match="UNSUPPORTED|status -3"):
        ops.conv_igemm16(torch.randn(2, 8, 8, 48, device="cuda").bfloat16(), torch.randn(64, 9 * 48, device="cuda").bfloat16(), 3, 1, 1)


def test_conv_bn_act16_node_implicit_vs_materialised(ops):
    """functional16.ConvBNAct16Fn (bf16 storage and the fp32-held form) with and without the implicit kernel: outputs equal up to rare
    one-ulp flips of the conv result, gradients within the same flips' reach"""
    from dinov2_distillation_amd import functional as Fn, functional16 as F16
    for io16 in (True, False):
        for (B, H, C, O, k, s) in [(8, 14, 256, 256, 3, 1), (4, 28, 64, 128, 3, 2)]:
            g = torch.Generator().manual_seed(11 * H + O)
            x0 = torch.randn(B, H, H, C, generator=g).bfloat16()
            x0 = x0.cuda() if io16 else x0.float().cuda()
            w0 = (torch.randn(O, C, k, k, generator=g) / (k * C ** 0.5)).cuda()
            gam, bet = (1 + 0.1 * torch.randn(O, generator=g)).cuda(), (0.1 * torch.randn(O, generator=g)).cuda()
            outs = []
            for flag in (True, False):
                Fn.CONV_IGEMM = flag
                try:
                    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
                    gw, gb = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
                    state = (torch.zeros(O).cuda(), torch.ones(O).cuda(), 0.1, 1e-5)
                    y = F16.ConvBNAct16Fn.apply(x, w, gw, gb, state, k, s, 1, True, False)
                    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(y.dtype).cuda()
                    y.backward(dy)
                    torch.cuda.synchronize()
                    outs.append((y.detach().float(), x.grad.float(), w.grad, gw.grad, gb.grad))
                finally:
                    Fn.CONV_IGEMM = True
            for a, b, what in zip(outs[0], outs[1], ("y", "dx", "dw", "dgamma", "dbeta")):
                close(a, b, f"{what} io16={io16} [{B},{H},{C}->{O},s{s}]", rtol=2e-2 if what in ("y", "dx") else 5e-3)
                assert (a - b).abs().mean().item() <= 2e-4 * b.abs().mean().item() + 1e-12, f"{what}: not just rare flips"
