// General (overlapping) convolutions of the STDC / ResNet students as im2col + the fp32 MFMA GEMM, plus the small depthwise
// 3x3 family (STDC "avd" layer; AvgPool2d(3,2,1) is the same kernel with all taps = 1/9), all on NHWC fp32:
//   models/backbones/stdc.py:8-24 (ConvX: conv k in {1,3}, stride {1,2}, no bias), :113-126 (avd depthwise 3x3 s2, AvgPool skip)
// Every adjoint is written in GATHER form (each input element sums the outputs that touched it): deterministic, no atomics.
#include "common.h"
#include "dkd_hip.h"

namespace {

inline int blocks_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// rows[(b,oy,ox)][(ky,kx,c)] = x[b, oy*s+ky-p, ox*s+kx-p, c]  (0 outside); C % 4 == 0
__global__ void im2col_nhwc_kernel(const float4* __restrict__ x, float4* __restrict__ rows, int B, int H, int W, int C,
                                   int k, int s, int p, int Ho, int Wo) {
    const int cq = C >> 2;
    const long total = (long)B * Ho * Wo * k * k * cq;
    GRID_STRIDE(i, total) {
        long r = i;
        const int c4 = (int)(r % cq); r /= cq;
        const int kx = (int)(r % k); r /= k;
        const int ky = (int)(r % k); r /= k;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int y = oy * s + ky - p, xx = ox * s + kx - p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && xx >= 0 && xx < W) v = x[(((long)b * H + y) * W + xx) * cq + c4];
        rows[i] = v;
    }
}

// adjoint of the above: dx[b,y,x,c] = sum_{ky,kx} drows[(b,(y+p-ky)/s,(x+p-kx)/s)][(ky,kx,c)] over valid, divisible taps
__global__ void col2im_nhwc_kernel(const float4* __restrict__ drows, float4* __restrict__ dx, int B, int H, int W, int C,
                                   int k, int s, int p, int Ho, int Wo) {
    const int cq = C >> 2;
    const long total = (long)B * H * W * cq;
    GRID_STRIDE(i, total) {
        long r = i;
        const int c4 = (int)(r % cq); r /= cq;
        const int xx = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < k; ++ky) {
            const int ty = y + p - ky;
            if (ty < 0 || ty % s) continue;
            const int oy = ty / s;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = xx + p - kx;
                if (tx < 0 || tx % s) continue;
                const int ox = tx / s;
                if (ox >= Wo) continue;
                const float4 v = drows[((((long)b * Ho + oy) * Wo + ox) * k * k + ky * k + kx) * cq + c4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        dx[i] = acc;
    }
}

// first layer: NCHW image -> rows[(b,oy,ox)][(c,ky,kx) padded to Kp]  (matches weight [O][c][ky][kx]; pad columns are 0)
__global__ void im2col_nchw_kernel(const float* __restrict__ img, float* __restrict__ rows, int B, int Cin, int H, int W,
                                   int k, int s, int p, int Ho, int Wo, int Kp) {
    const long total = (long)B * Ho * Wo * Kp;
    const int K = Cin * k * k;
    GRID_STRIDE(i, total) {
        const int kk = (int)(i % Kp);
        long r = i / Kp;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v = 0.f;
        if (kk < K) {
            const int kx = kk % k, ky = (kk / k) % k, c = kk / (k * k);
            const int y = oy * s + ky - p, xx = ox * s + kx - p;
            if (y >= 0 && y < H && xx >= 0 && xx < W) v = img[(((long)b * Cin + c) * H + y) * W + xx];
        }
        rows[i] = v;
    }
}

// depthwise 3x3, dilation d, pad d, stride s, tap-major weights [9][C] (nullptr => all taps 1/9 == AvgPool2d(3, s, 1),
// count_include_pad).  Output extent (H - 1) / s + 1 for every dilation.
template <int K>
__global__ void dwconv_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ wt, float4* __restrict__ out,
                                   int B, int H, int W, int C, int s, int Ho, int Wo, int d) {
    const int cq = C >> 2;
    const long total = (long)B * Ho * Wo * cq;
    const float ninth = 1.f / (float)(K * K);
    GRID_STRIDE(i, total) {
        long r = i;
        const int c4 = (int)(r % cq); r /= cq;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int y = oy * s + (ky - K / 2) * d;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xx = ox * s + (kx - K / 2) * d;
                if (xx < 0 || xx >= W) continue;
                const float4 v = x[(((long)b * H + y) * W + xx) * cq + c4];
                const float4 w = wt ? wt[(ky * K + kx) * cq + c4] : make_float4(ninth, ninth, ninth, ninth);
                acc.x = fmaf(w.x, v.x, acc.x); acc.y = fmaf(w.y, v.y, acc.y); acc.z = fmaf(w.z, v.z, acc.z); acc.w = fmaf(w.w, v.w, acc.w);
            }
        }
        out[i] = acc;
    }
}

template <int K>
__global__ void dwconv_dgrad_kernel(const float4* __restrict__ dout, const float4* __restrict__ wt, float4* __restrict__ dx,
                                     int B, int H, int W, int C, int s, int Ho, int Wo, int d) {
    const int cq = C >> 2;
    const long total = (long)B * H * W * cq;
    const float ninth = 1.f / (float)(K * K);
    GRID_STRIDE(i, total) {
        long r = i;
        const int c4 = (int)(r % cq); r /= cq;
        const int xx = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ty = y + (K / 2 - ky) * d;
            if (ty < 0 || ty % s) continue;
            const int oy = ty / s;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int tx = xx + (K / 2 - kx) * d;
                if (tx < 0 || tx % s) continue;
                const int ox = tx / s;
                if (ox >= Wo) continue;
                const float4 g = dout[(((long)b * Ho + oy) * Wo + ox) * cq + c4];
                const float4 w = wt ? wt[(ky * K + kx) * cq + c4] : make_float4(ninth, ninth, ninth, ninth);
                acc.x = fmaf(w.x, g.x, acc.x); acc.y = fmaf(w.y, g.y, acc.y); acc.z = fmaf(w.z, g.z, acc.z); acc.w = fmaf(w.w, g.w, acc.w);
            }
        }
        dx[i] = acc;
    }
}

// partial weight gradients: thread = (channel quad, image, output row); part[(b*Ho + oy)][tap][c]
template <int K>
__global__ void dwconv_wgrad_kernel(const float4* __restrict__ x, const float4* __restrict__ dout, float4* __restrict__ part,
                                     int B, int H, int W, int C, int s, int Ho, int Wo, int d) {
    const int cq = C >> 2;
    const long total = (long)B * Ho * cq;
    GRID_STRIDE(i, total) {
        const int c4 = (int)(i % cq);
        const long r = i / cq;
        const int oy = (int)(r % Ho), b = (int)(r / Ho);
        float4 acc[K * K];
#pragma unroll
        for (int t = 0; t < K * K; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ox = 0; ox < Wo; ++ox) {
            const float4 g = dout[(((long)b * Ho + oy) * Wo + ox) * cq + c4];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int y = oy * s + (ky - K / 2) * d;
                if (y < 0 || y >= H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int xx = ox * s + (kx - K / 2) * d;
                    if (xx < 0 || xx >= W) continue;
                    const float4 v = x[(((long)b * H + y) * W + xx) * cq + c4];
                    float4& a = acc[ky * K + kx];
                    a.x = fmaf(g.x, v.x, a.x); a.y = fmaf(g.y, v.y, a.y); a.z = fmaf(g.z, v.z, a.z); a.w = fmaf(g.w, v.w, a.w);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < K * K; ++t) part[(r * K * K + t) * cq + c4] = acc[t];
    }
}

// F.max_pool2d(kernel 3, stride 2, pad 1) (resnet.py:543): first maximum in row-major window order, its flat input index
// kept for the backward pass
__global__ void maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int* __restrict__ arg, int B, int H,
                                    int W, int C, int s, int Ho, int Wo) {
    const long total = (long)B * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        long r = i;
        const int c = (int)(r % C); r /= C;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float best = -INFINITY;
        int bi = -1;
        for (int ky = 0; ky < 3; ++ky) {
            const int y = oy * s + ky - 1;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = ox * s + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const float v = x[(((long)b * H + y) * W + xx) * C + c];
                if (v > best || bi < 0) { best = v; bi = y * W + xx; }
            }
        }
        out[i] = best;
        arg[i] = bi;
    }
}
__global__ void maxpool3_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg, float* __restrict__ dx, int B,
                                    int H, int W, int C, int s, int Ho, int Wo) {
    const long total = (long)B * H * W * C;
    GRID_STRIDE(i, total) {
        long r = i;
        const int c = (int)(r % C); r /= C;
        const int xx = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || ty % s) continue;
            const int oy = ty / s;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = xx + 1 - kx;
                if (tx < 0 || tx % s) continue;
                const int ox = tx / s;
                if (ox >= Wo) continue;
                const long o = (((long)b * Ho + oy) * Wo + ox) * C + c;
                if (arg[o] == y * W + xx) acc += dout[o];
            }
        }
        dx[i] = acc;
    }
}
__global__ void add_relu_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long nvec) {
    GRID_STRIDE(i, nvec) {
        const float4 u = a[i], v = b[i];
        out[i] = make_float4(fmaxf(u.x + v.x, 0.f), fmaxf(u.y + v.y, 0.f), fmaxf(u.z + v.z, 0.f), fmaxf(u.w + v.w, 0.f));
    }
}

// dst[r*ldd + c] (+)= src[r*lds + c], c < cols: channel concat / slice on token-major tensors
__global__ void copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long rows, int cols,
                              int accumulate) {
    const int cq = cols >> 2;
    const long total = rows * cq;
    GRID_STRIDE(i, total) {
        const long r = i / cq;
        const int c4 = (int)(i % cq);
        float4 v = *reinterpret_cast<const float4*>(src + r * lds + 4 * c4);
        float4* d = reinterpret_cast<float4*>(dst + r * ldd + 4 * c4);
        if (accumulate) {
            const float4 o = *d;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *d = v;
    }
}
// nn.AvgPool2d(2, 2, 0, ceil_mode=True) on NHWC (models/backbones/presnet.py:227,268: the "d" shortcut): windows hanging over
// the bottom / right edge are clipped and divided by the number of in-bounds elements.  mode 1 = backward (in = dy [Ho,Wo]).
__global__ void avgpool2_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H, int W, int cq, int Ho,
                                int Wo, int backward) {
    const long total = (long)B * (backward ? H * W : Ho * Wo) * cq;
    GRID_STRIDE(i, total) {
        long r = i;
        const int c = (int)(r % cq); r /= cq;
        if (!backward) {
            const int ox = (int)(r % Wo); r /= Wo;
            const int oy = (int)(r % Ho);
            const int b = (int)(r / Ho);
            const int ny = min(2, H - 2 * oy), nx = min(2, W - 2 * ox);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ky = 0; ky < ny; ++ky)
                for (int kx = 0; kx < nx; ++kx) {
                    const float4 v = in[(((long)b * H + 2 * oy + ky) * W + 2 * ox + kx) * cq + c];
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
            const float inv = 1.f / (float)(ny * nx);
            out[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        } else {
            const int x = (int)(r % W); r /= W;
            const int y = (int)(r % H);
            const int b = (int)(r / H);
            const int oy = y >> 1, ox = x >> 1;
            const float inv = 1.f / (float)(min(2, H - 2 * oy) * min(2, W - 2 * ox));
            const float4 v = in[(((long)b * Ho + oy) * Wo + ox) * cq + c];
            out[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        }
    }
}

}  // namespace

#define F4(p) reinterpret_cast<const float4*>(p)
#define F4W(p) reinterpret_cast<float4*>(p)

extern "C" int dkd_im2col_nhwc_f32(const float* x, float* rows, int B, int H, int W, int C, int k, int stride, int pad,
                                   int inverse, void* stream) {
    if (!x || !rows || B <= 0 || k <= 0 || stride <= 0) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_ALIGN;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    if (!inverse) {
        const long total = (long)B * Ho * Wo * k * k * (C / 4);
        hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, st, F4(x), F4W(rows), B, H, W, C, k,
                           stride, pad, Ho, Wo);
    } else {  // x is the OUTPUT (dx), rows the input (drows)
        const long total = (long)B * H * W * (C / 4);
        hipLaunchKernelGGL(col2im_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, st, F4(rows), F4W(const_cast<float*>(x)),
                           B, H, W, C, k, stride, pad, Ho, Wo);
    }
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_im2col_nchw_f32(const float* img, float* rows, int B, int Cin, int H, int W, int k, int stride, int pad,
                                   int Kp, void* stream) {
    if (!img || !rows || B <= 0 || k <= 0 || stride <= 0 || Kp < Cin * k * k) return DKD_ERR_BAD_ARG;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long total = (long)B * Ho * Wo * Kp;
    hipLaunchKernelGGL(im2col_nchw_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, img, rows, B, Cin, H,
                       W, k, stride, pad, Ho, Wo, Kp);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dwconvk_f32(const float* in, const float* w_tap_major, float* out, int B, int H, int W, int C, int k, int stride,
                               int dilation, int mode, void* stream) {
    // depthwise k x k (k = 3 or 5), padding (k / 2) * dilation.  mode 0: forward (in = x [B,H,W,C], out = y [B,Ho,Wo,C]);
    // mode 1: dgrad (in = dy [B,Ho,Wo,C], out = dx [B,H,W,C]).  w_tap_major [k*k, C] (NULL with k = 3: the 1/9 average pool)
    if (!in || !out || B <= 0 || stride <= 0 || dilation <= 0) return DKD_ERR_BAD_ARG;
    if (k != 3 && k != 5) return DKD_ERR_UNSUPPORTED;
    if (C & 3) return DKD_ERR_ALIGN;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;  // (H + 2p - d(k-1) - 1) / s + 1 with p = (k/2) d
    hipStream_t st = (hipStream_t)stream;
    const long total = mode == 0 ? (long)B * Ho * Wo * (C / 4) : (long)B * H * W * (C / 4);
#define DKD_DW(KK)                                                                                                             \
    if (mode == 0)                                                                                                             \
        hipLaunchKernelGGL(dwconv_fwd_kernel<KK>, dim3(blocks_for(total)), dim3(256), 0, st, F4(in), F4(w_tap_major), F4W(out), \
                           B, H, W, C, stride, Ho, Wo, dilation);                                                              \
    else                                                                                                                       \
        hipLaunchKernelGGL(dwconv_dgrad_kernel<KK>, dim3(blocks_for(total)), dim3(256), 0, st, F4(in), F4(w_tap_major),        \
                           F4W(out), B, H, W, C, stride, Ho, Wo, dilation)
    if (k == 3) { DKD_DW(3); } else { DKD_DW(5); }
#undef DKD_DW
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dwconvk_wgrad_f32(const float* x, const float* dout, float* partials, int B, int H, int W, int C, int k,
                                     int stride, int dilation, void* stream) {
    if (!x || !dout || !partials || B <= 0 || stride <= 0 || dilation <= 0) return DKD_ERR_BAD_ARG;
    if (k != 3 && k != 5) return DKD_ERR_UNSUPPORTED;
    if (C & 3) return DKD_ERR_ALIGN;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long total = (long)B * Ho * (C / 4);
    if (k == 3)
        hipLaunchKernelGGL(dwconv_wgrad_kernel<3>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, F4(x), F4(dout),
                           F4W(partials), B, H, W, C, stride, Ho, Wo, dilation);
    else
        hipLaunchKernelGGL(dwconv_wgrad_kernel<5>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, F4(x), F4(dout),
                           F4W(partials), B, H, W, C, stride, Ho, Wo, dilation);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dwconv3_f32(const float* in, const float* w_tap_major, float* out, int B, int H, int W, int C, int stride,
                               int dilation, int mode, void* stream) {
    return dkd_dwconvk_f32(in, w_tap_major, out, B, H, W, C, 3, stride, dilation, mode, stream);
}

extern "C" int dkd_dwconv3_wgrad_f32(const float* x, const float* dout, float* partials, int B, int H, int W, int C,
                                     int stride, int dilation, void* stream) {
    return dkd_dwconvk_wgrad_f32(x, dout, partials, B, H, W, C, 3, stride, dilation, stream);
}

extern "C" int dkd_copy2d_f32(const float* src, long lds, float* dst, long ldd, long rows, int cols, int accumulate,
                              void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return DKD_ERR_BAD_ARG;
    if ((cols & 3) || (lds & 3) || (ldd & 3) || !dkd_aligned16(src) || !dkd_aligned16(dst)) return DKD_ERR_ALIGN;
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for(rows * (cols / 4))), dim3(256), 0, (hipStream_t)stream, src, lds, dst,
                       ldd, rows, cols, accumulate);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_maxpool3_f32(const float* in, float* out, int* argmax, int B, int H, int W, int C, int stride, int mode,
                                void* stream) {
    // mode 0: forward (in = x, out = y, argmax written); mode 1: backward (in = dy, out = dx, argmax read)
    if (!in || !out || !argmax || B <= 0 || stride <= 0) return DKD_ERR_BAD_ARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0)
        hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(blocks_for((long)B * Ho * Wo * C)), dim3(256), 0, st, in, out, argmax, B,
                           H, W, C, stride, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(blocks_for((long)B * H * W * C)), dim3(256), 0, st, in, argmax, out, B, H,
                           W, C, stride, Ho, Wo);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_avgpool2_f32(const float* in, float* out, int B, int H, int W, int C, int mode, void* stream) {
    // mode 0: forward (in = x [B,H,W,C], out = y [B,ceil(H/2),ceil(W/2),C]); mode 1: backward (in = dy, out = dx [B,H,W,C])
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_ALIGN;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)B * (mode ? H * W : Ho * Wo) * (C / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), B, H, W, C / 4, Ho, Wo, mode);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_add_relu_f32(const float* a, const float* b, float* out, long n, void* stream) {
    if (!a || !b || !out || n <= 0) return DKD_ERR_BAD_ARG;
    if (n & 3) return DKD_ERR_ALIGN;
    hipLaunchKernelGGL(add_relu_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, F4(a), F4(b), F4W(out),
                       n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
