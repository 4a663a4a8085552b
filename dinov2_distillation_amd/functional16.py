"""Autograd nodes of the MIXED-PRECISION student: the ConvNeXt-V2 backbone under the reference's default
``precision=16`` policy (/root/reference/train.py:263 == torch autocast; bf16 instead of fp16 on MI355X: native MFMA rate, fp32's
exponent range, no loss scaling).  What autocast keeps in 16 bit lives in HBM as bf16 (conv / Linear inputs and outputs, GELU
output, the residual stream of stages 1-3), what it computes in fp32 stays fp32 (LayerNorm, GRN's norm and affine arithmetic,
parameters, statistics, the residual stream of stage 0 - the stem's channels-first LayerNorm returns fp32).  Rounding points, per
block (models/backbones/convnext.py:75-87 under autocast):

    d16 = rnd(dwconv7(x16; rnd(W), rnd(b)))          conv2d: every tensor argument cast to 16 bit, 16-bit result
    n16 = rnd(LayerNorm_fp32(d16))                   layer_norm runs in fp32; the next Linear casts its input
    h16 = rnd(n16 W1^T + b1)                         W1, b1 cast to 16 bit
    g16 = rnd(gelu(h16))                             (recomputed from h16 wherever it is needed, never stored)
    y16 = rnd(gamma (g16 Nx) + beta + g16)           norm / mean / affine in fp32; the next Linear casts its input
    out = x + rnd(y16 W2^T + b2)                     rounded again when x is 16-bit

and the mirror image in the backward pass (every gradient of a 16-bit tensor is a 16-bit tensor; weight gradients are the 16-bit
results of the backward matmuls, cast up into the fp32 master gradients).  All arithmetic inside the kernels is fp32."""
import torch
from torch.autograd import Function

from . import ops
from .functional import _c, direct, ret

BF16, F32 = torch.bfloat16, torch.float32


def _w16(w):
    """a parameter as the 16-bit operand autocast makes of it"""
    return ops.cast_bf16(w.reshape(-1, w.shape[-1]) if w.dim() > 1 else w.reshape(1, -1)).reshape(w.shape)


def _rounded(t):
    """fp32 copy holding bf16 values (taps / biases that a kernel consumes as fp32)"""
    return ops.round_bf16_(t.detach().clone())


def wgrad16(dy16, x16, w, b, transform=None, reshape=None):
    """dW = rnd(dy16^T x16) and db = rnd(sum_rows dy16), as fp32 master gradients.  In-place on the weight-gradient stream when
    the arena allows it (ops.can_async), else returned to autograd."""
    N, K = dy16.shape[-1], x16.shape[-1]
    if ops.can_async(w, b):
        side = ops.wgrad_stream(dy16.device)
        ops.claim(w, b)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if transform is None:
                ops.gemm_bf16(dy16, x16, a_layout=1, b_layout=1, epi=ops.EPI_F32, out=w.grad.view(N, K), ws_slot=3)
            else:
                transform(ops.gemm_bf16(dy16, x16, a_layout=1, b_layout=1, epi=ops.EPI_F32, ws_slot=3))
            if b is not None:
                ops.round_bf16_(ops.colreduce16(dy16, ops.RED_SUM, out=b.grad.view(1, -1)))
        dy16.record_stream(side)
        x16.record_stream(side)
        ops.grad_done(w, b)
        return None, None
    dw = ops.gemm_bf16(dy16, x16, a_layout=1, b_layout=1, epi=ops.EPI_F32)
    if reshape is not None:
        dw = reshape(dw)
    db = None if b is None else ops.round_bf16_(ops.colreduce16(dy16, ops.RED_SUM)).reshape(-1)
    return dw, db


class ToFloatFn(Function):
    """16-bit feature -> fp32 (what an fp32 consumer of an autocast tensor sees); the gradient comes back rounded to 16 bit"""

    @staticmethod
    def forward(ctx, x16):
        return ops.cast_f32(_c(x16))

    @staticmethod
    def backward(ctx, dy):
        return ops.cast_bf16(_c(dy)).reshape(dy.shape)


class LayerNorm16Fn(Function):
    """The reference's hand-written channels-first LayerNorm (convnext.py:31-36; stem and downsample layers) on NHWC rows under
    autocast: on an fp32 input it is plain fp32 arithmetic; on a 16-bit input the mean and the centred values are 16-bit
    results, pow / mean / sqrt / affine fp32 (cf16 form of the kernels).  fp32 or bf16 output (out16: the rounding the next
    conv applies to its input)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out16):
        x = _c(x)
        cf16 = x.dtype == BF16
        y, mean, rstd = ops.layernorm_fwd_io(x, gamma, beta, eps, BF16 if out16 else F32, cf16=cf16)
        ctx.save_for_backward(x, mean, rstd, gamma)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        og, ob = direct(*ctx.params)
        dx, dg, db = ops.layernorm_bwd_io(_c(dy), x, mean, rstd, gamma, x.dtype, dg_out=og, db_out=ob, cf16=x.dtype == BF16)
        return dx, ret(og, dg, ctx.params[0]), ret(ob, db, ctx.params[1]), None, None


class PatchConv16Fn(Function):
    """k x k stride-k convolution under autocast: patchify + 16-bit GEMM, 16-bit output.  src_nchw: the fp32 NCHW image (stem,
    convnext.py:117; no input gradient); otherwise a bf16 NHWC feature (downsample conv, :124)."""

    @staticmethod
    def forward(ctx, x, w, b, k, src_nchw):
        O = w.shape[0]
        x = _c(x)
        if src_nchw:
            B, _, H, W = x.shape
            rows = ops.cast_bf16(ops.patchify_nchw(x, k))
            w2 = ops.cast_bf16(w.reshape(O, -1))
        else:
            B, H, W, C = x.shape
            # pure data movement: a bf16 [.., C] tensor is an fp32 [.., C / 2] tensor to the patchify kernel
            rows = ops.patchify_nhwc(x.view(F32), k).view(BF16)
            w2 = ops.cast_bf16(ops.transpose(w.reshape(O, C, k * k)).reshape(O, k * k * C))  # [O][ky][kx][c]
        y = ops.gemm_bf16(rows, w2, bias=_w16(b))
        ctx.save_for_backward(rows, w2)
        ctx.meta = (k, src_nchw, tuple(x.shape), tuple(w.shape))
        ctx.params = (w, b)
        return y.reshape(B, H // k, W // k, O)

    @staticmethod
    def backward(ctx, dy):
        rows, w2 = ctx.saved_tensors
        k, src_nchw, xshape, wshape = ctx.meta
        O = wshape[0]
        dy2 = _c(dy).reshape(-1, O)
        w, b = ctx.params
        dx = None
        if src_nchw:
            dw, db = wgrad16(dy2, rows, w, b, reshape=lambda t: t.reshape(wshape))
        else:
            B, H, W, C = xshape
            dw, db = wgrad16(dy2, rows, w, b,
                             transform=lambda t: ops.transpose(t.reshape(O, k * k, C), out=w.grad.view(O, C, k * k)),
                             reshape=lambda t: ops.transpose(t.reshape(O, k * k, C)).reshape(wshape))
            if ctx.needs_input_grad[0]:
                drows = ops.gemm_bf16(dy2, w2, b_layout=1)
                dx = ops.unpatchify_nhwc(drows.view(F32), B, H, W, C // 2, k).view(BF16)
        return dx, dw, db, None, None


class ConvNeXtBlock16Fn(Function):
    """ConvNeXt-V2 block (convnext.py:75-87) under autocast, NHWC, bf16 activations in HBM (see the module docstring).
    x bf16 (stages 1-3) or fp32 (stage 0: the residual stream stays fp32 there)."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_w, ln_b, w1, b1, grn_g, grn_b, w2, b2):
        x = _c(x)
        B, H, W, C = x.shape
        M = B * H * W
        x16 = x if x.dtype == BF16 else ops.cast_bf16(x).reshape(x.shape)
        wt = ops.round_bf16_(ops.transpose(dw_w.reshape(C, 49)))  # tap-major [49, C], 16-bit values
        d16 = ops.dwconv7_io(x16, wt, _rounded(dw_b))
        n16, mean, rstd = ops.layernorm_fwd_io(d16, ln_w, ln_b, 1e-6, BF16)
        w1h, w2h = ops.cast_bf16(w1), ops.cast_bf16(w2)
        h16 = ops.gemm_bf16(n16.reshape(M, C), w1h, bias=_w16(b1))
        gg, gb = grn_g.reshape(-1), grn_b.reshape(-1)
        y16, stats = ops.grn_fwd16(h16, gg, gb, B)
        if x.dtype == BF16:
            out = ops.gemm_bf16_resid16(y16, w2h, _w16(b2), x.reshape(M, C))
        else:
            out = ops.gemm_bf16(y16, w2h, epi=ops.EPI_F32_RESID, bias=_w16(b2), residual=x.reshape(M, C))
        ctx.save_for_backward(x16, wt, d16, mean, rstd, n16, h16, y16, w1h, w2h, ln_w, gg, *stats)
        ctx.shapes = (B, H, W, C, tuple(dw_w.shape), tuple(grn_g.shape))
        ctx.params = (w1, b1, w2, b2)
        ctx.small = (dw_w, dw_b, ln_w, ln_b, grn_g, grn_b)
        return out.reshape(B, H, W, C)

    @staticmethod
    def backward(ctx, dout):
        x16, wt, d16, mean, rstd, n16, h16, y16, w1h, w2h, ln_w, gg, gx, nx, denom = ctx.saved_tensors
        B, H, W, C, dw_shape, grn_shape = ctx.shapes
        M = B * H * W
        dout = _c(dout)
        do16 = (dout if dout.dtype == BF16 else ops.cast_bf16(dout)).reshape(M, C)
        pw1, pb1, pw2, pb2 = ctx.params
        dw2, db2 = wgrad16(do16, y16, pw2, pb2)
        dy16 = ops.gemm_bf16(do16, w2h, b_layout=1)  # [M, 4C]
        p_dww, p_dwb, p_lnw, p_lnb, p_gg, p_gb = ctx.small
        o_dww, o_dwb, o_lnw, o_lnb, o_gg, o_gb = direct(*ctx.small)
        dh16, dgrn_g, dgrn_b = ops.grn_gelu_bwd16(dy16, h16, (gx, nx, denom), gg, B,
                                                  None if o_gg is None else o_gg.view(-1), None if o_gb is None else o_gb.view(-1))
        dw1, db1 = wgrad16(dh16, n16.reshape(M, C), pw1, pb1)
        dn16 = ops.gemm_bf16(dh16, w1h, b_layout=1)  # [M, C]
        dd16, dln_w, dln_b = ops.layernorm_bwd_io(dn16, d16.reshape(M, C), mean, rstd, ln_w, BF16, dg_out=o_lnw, db_out=o_lnb)
        dd16 = dd16.reshape(B, H, W, C)
        dx = ops.dwconv7_io(dd16, wt, None, add=dout.reshape(B, H, W, C), flip=True, out_dtype=dout.dtype)
        ddw = ops.round_bf16_(ops.transpose(ops.dwconv7_wgrad16(x16, dd16), out=None if o_dww is None else o_dww.view(C, 49)))
        ddb = ops.round_bf16_(ops.colreduce16(dd16.reshape(M, C), ops.RED_SUM, out=None if o_dwb is None else o_dwb.view(1, C)))
        return (dx, ret(o_dww, ddw.reshape(dw_shape), p_dww), ret(o_dwb, ddb.reshape(-1), p_dwb), ret(o_lnw, dln_w, p_lnw),
                ret(o_lnb, dln_b, p_lnb), dw1, db1, ret(o_gg, dgrn_g.reshape(grn_shape), p_gg),
                ret(o_gb, dgrn_b.reshape(grn_shape), p_gb), dw2, db2)
