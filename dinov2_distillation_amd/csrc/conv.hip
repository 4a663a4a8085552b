// Spatial kernels of the ConvNeXt-V2 student and of the feature plumbing, all on NHWC (token-major) fp32 tensors:
//   * depthwise 7x7 (pad 3) forward / dgrad (same kernel, flipped taps) / wgrad   (models/backbones/convnext.py:62-64)
//   * patchify / unpatchify for the non-overlapping convs (stem 4x4 s4, downsample 2x2 s2, DINOv2 patch-embed 14x14
//     s14), which turn those convs into plain GEMMs                                (convnext.py:116-126)
//   * bilinear resize, align_corners=False, forward and (gather-form, deterministic) backward
//                                                                                  (models/model_zoo.py:123-128)
// Channel is the fastest index, so consecutive lanes read consecutive channels: every access is a dense 16 B/lane run.
#include <cstdlib>
#include "common.h"
#include "dkd_hip.h"

namespace {

__device__ __forceinline__ void fma4(float4& a, const float4 w, const float4 x) {
    a.x = fmaf(w.x, x.x, a.x); a.y = fmaf(w.y, x.y, a.y); a.z = fmaf(w.z, x.z, a.z); a.w = fmaf(w.w, x.w, a.w);
}

// thread = (channel quad, 7 consecutive x, one y, one image).  wt is tap-major [49][C].
// The kernel is VALU-bound (98 packed FMAs per input row), so everything around the FMAs is kept off the vector ALU:
// addresses are 32-bit float4 indices off the wave-uniform tensor base (one add per load instead of a 64-bit multiply-add
// chain; the launcher checks the tensor has < 2^32 float4), out-of-range columns are loaded from the clamped column and
// zeroed afterwards only in the threads of the first / last strip of a row.
// TI / TA / TO: element types of the input, of the optional addend and of the output (float or __bf16, common.h).  A bf16
// output is rounded once after the taps (the convolution's own 16-bit result) and again after the addend (the 16-bit sum
// autograd forms when it accumulates the two gradients), as the autocast step does.
template <bool FLIP, typename TI = float, typename TA = float, typename TO = float>
__global__ __launch_bounds__(256) void dwconv7_kernel(const TI* __restrict__ in, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, const TA* __restrict__ add,
                                                      TO* __restrict__ out, int B, int H, int W, int C) {
    const unsigned cq = C >> 2, xt = (W + 6) / 7;
    const unsigned total = (unsigned)B * H * xt * cq;  // < 2^32 (launcher): the index split below stays in 32-bit division
    const float4* w4 = reinterpret_cast<const float4*>(wt);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned c4 = idx % cq;
        unsigned r = idx / cq;
        const int x0 = (int)(r % xt) * 7;
        r /= xt;
        const int y = (int)(r % (unsigned)H), b = (int)(r / (unsigned)H);
        float4 acc[7];
        const float4 bv = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] = bv;
        const bool edge = x0 < 3 || x0 + 9 >= W;
        unsigned xo[13];  // float4 offset of column x0 + k - 3 (clamped) within a row
#pragma unroll
        for (int k = 0; k < 13; ++k) xo[k] = (unsigned)min(max(x0 + k - 3, 0), W - 1) * (unsigned)cq + c4;
#pragma unroll 1
        for (int i = 0; i < 7; ++i) {
            const int yy = y + i - 3;
            if (yy < 0 || yy >= H) continue;
            const unsigned rowb = (unsigned)((b * H + yy) * W) * (unsigned)cq;
            float4 v[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) v[k] = ld4(in, rowb + xo[k]);
            if (edge) {
#pragma unroll
                for (int k = 0; k < 13; ++k)
                    if (x0 + k - 3 < 0 || x0 + k - 3 >= W) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const unsigned wb = (unsigned)(FLIP ? 48 - i * 7 : i * 7) * (unsigned)cq + c4;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float4 wv = FLIP ? w4[wb - (unsigned)j * cq] : w4[wb + (unsigned)j * cq];
#pragma unroll
                for (int k = 0; k < 7; ++k) fma4(acc[k], wv, v[k + j]);
            }
        }
        const unsigned ob = (unsigned)((b * H + y) * W + x0) * (unsigned)cq + c4;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (x0 + k >= W) break;
            float4 o = acc[k];
            if (add) {
                if (dkd_is16<TI>::value) o = rnd16(o);
                const float4 a = ld4(add, ob + (unsigned)k * cq);
                o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            }
            st4(out, ob + (unsigned)k * cq, o);
        }
    }
}

// Second form of the same convolution, for when the weights of a workgroup's channel range fit in LDS (always, for the
// ConvNeXt widths): the first form is bound by the CU's vector-memory path (13 input + 7 weight float4 loads per 98 packed
// FMAs: 20 x 1 KB per wave against 64 B/clk), not by HBM and not by the VALU.  Here
//   * the 49 x CB weight float4 of the block's CB channel quads are staged once in LDS (ds_read_b128 has its own 128 B/clk),
//   * a thread produces RY output rows x 7 columns: an input row is loaded once and feeds up to RY tap rows (vertical
//     reuse: (RY + 6) / RY input rows per output row instead of 7).
// block = CB channel quads x SP spatial tiles (CB * SP <= 256), grid.y = channel chunks, grid.x strides over the tiles.
template <bool FLIP, int RY, int XW, typename TI = float, typename TA = float, typename TO = float>
__global__ __launch_bounds__(256) void dwconv7_lds_kernel(const TI* __restrict__ in, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, const TA* __restrict__ add,
                                                          TO* __restrict__ out, int B, int H, int W, int C, int CB, int SP) {
    extern __shared__ float4 wl[];  // [49][CB], already flipped for the dgrad form
    const unsigned cq = C >> 2, xt = (W + XW - 1) / XW, nyb = (H + RY - 1) / RY;
    const unsigned c40 = blockIdx.y * CB;
    const float4* w4 = reinterpret_cast<const float4*>(wt);
    for (unsigned t = threadIdx.x; t < 49u * CB; t += 256) {
        const unsigned tap = t / CB, cc = t % CB;
        wl[t] = w4[(FLIP ? 48 - tap : tap) * cq + c40 + cc];
    }
    __syncthreads();
    const unsigned cc = threadIdx.x % CB, sp = threadIdx.x / CB;
    if (sp >= (unsigned)SP) return;
    const unsigned c4 = c40 + cc;
    const unsigned total = (unsigned)B * nyb * xt;
    const float4 bv = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned st = blockIdx.x * SP + sp; st < total; st += gridDim.x * SP) {
        const int x0 = (int)(st % xt) * XW;
        unsigned r = st / xt;
        const int y0 = (int)(r % nyb) * RY, b = (int)(r / nyb);
        float4 acc[RY][XW];
#pragma unroll
        for (int ro = 0; ro < RY; ++ro)
#pragma unroll
            for (int k = 0; k < XW; ++k) acc[ro][k] = bv;
        const bool edge = x0 < 3 || x0 + XW + 2 >= W;
        unsigned xo[XW + 6];
#pragma unroll
        for (int k = 0; k < XW + 6; ++k) xo[k] = (unsigned)min(max(x0 + k - 3, 0), W - 1) * cq + c4;
#pragma unroll
        for (int t = 0; t < RY + 6; ++t) {  // fully unrolled: the (input row, output row) -> tap row map is static
            const int yy = y0 - 3 + t;
            if (yy < 0 || yy >= H) continue;
            const unsigned rowb = (unsigned)((b * H + yy) * W) * cq;
            float4 v[XW + 6];
#pragma unroll
            for (int k = 0; k < XW + 6; ++k) v[k] = ld4(in, rowb + xo[k]);
            if (edge) {
#pragma unroll
                for (int k = 0; k < XW + 6; ++k)
                    if (x0 + k - 3 < 0 || x0 + k - 3 >= W) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int ro = 0; ro < RY; ++ro) {
                const int i = t - ro;  // tap row of this input row for output row y0 + ro
                if (i < 0 || i > 6) continue;
                const float4* wr = wl + (unsigned)(i * 7) * CB + cc;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float4 wv = wr[(unsigned)j * CB];
#pragma unroll
                    for (int k = 0; k < XW; ++k) fma4(acc[ro][k], wv, v[k + j]);
                }
            }
        }
#pragma unroll
        for (int ro = 0; ro < RY; ++ro) {
            if (y0 + ro >= H) break;
            const unsigned ob = (unsigned)((b * H + y0 + ro) * W + x0) * cq + c4;
#pragma unroll
            for (int k = 0; k < XW; ++k) {
                if (x0 + k >= W) break;
                float4 o = acc[ro][k];
                if (add) {
                    if (dkd_is16<TI>::value) o = rnd16(o);
                    const float4 a = ld4(add, ob + (unsigned)k * cq);
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                st4(out, ob + (unsigned)k * cq, o);
            }
        }
    }
}

// wgrad partials: thread = (channel quad, tap row i, image, RY consecutive output rows); 7 float4 running sums, the input
// row slides through a 7-wide register window.  part[(b*ny + yc)][tap][c]
template <typename TI = float, typename TD = float>
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const TI* __restrict__ in, const TD* __restrict__ dout,
                                                            float* __restrict__ part, int B, int H, int W, int C,
                                                            int RY) {
    const int ny = (H + RY - 1) / RY, cq = C >> 2;
    const long total = (long)B * ny * 7 * cq;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cq);
        long r = idx / cq;
        const int i = (int)(r % 7);
        r /= 7;  // r = b * ny + yc
        const int yc = (int)(r % ny), b = (int)(r / ny);
        float4 acc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = zero;
        const long inb = (long)b * H * W * cq + c4, db = inb;  // 4-element index of (b, 0, 0, c4) in x and in dout
        const int y1 = min(H, (yc + 1) * RY);
        for (int y = yc * RY; y < y1; ++y) {
            const int yy = y + i - 3;
            if (yy < 0 || yy >= H) continue;
            const long irow = inb + (long)yy * W * cq;
            const long drow = db + (long)y * W * cq;
            float4 win[7];  // in[yy][x-3 .. x+3]
#pragma unroll
            for (int j = 0; j < 6; ++j) win[j + 1] = (j - 3 >= 0 && j - 3 < W) ? ld4(in, irow + (long)(j - 3) * cq) : zero;
            for (int x = 0; x < W; ++x) {
#pragma unroll
                for (int j = 0; j < 6; ++j) win[j] = win[j + 1];
                win[6] = (x + 3 < W) ? ld4(in, irow + (long)(x + 3) * cq) : zero;
                const float4 d = ld4(dout, drow + (long)x * cq);
#pragma unroll
                for (int j = 0; j < 7; ++j) fma4(acc[j], d, win[j]);
            }
        }
        float4* p = reinterpret_cast<float4*>(part + (r * 49 + i * 7) * C) + c4;
#pragma unroll
        for (int j = 0; j < 7; ++j) p[(long)j * cq] = acc[j];
    }
}

// wgrad, LDS-tiled: x and dy are read ONCE (the kernel above streams both once per tap row: 4.4 - 4.8x their bytes at the memory
// side, 0.06 - 0.15 of the HBM roofline).  One workgroup = one image, one chunk of 32 channels, one chunk of output rows; the input rows
// slide through a ring of eight LDS slots (a row is staged once - converted once if it is bf16 - and read by the seven tap rows that
// need it), the dy row through a double buffer; one barrier per row.  Lane = (tap row i, channel quad), wave = x segment: every
// thread keeps the 7 float4 sums of its tap row for its segment, the input row sliding through a 7-wide register window; the segments
// are added through LDS at the end.  Columns -3 .. -1 and W .. W+2 of every slot are zero for good, rows outside the image are staged as
// zeros.  Same partial layout as above: part[b * ny + yc][tap][c].
constexpr int DW_CC = 32;  // channels per workgroup (8 quads x 7 tap rows = 56 of a wave's 64 lanes)
template <typename TI, typename TD, int XS>
__global__ __launch_bounds__(XS * 64) void dwconv7_wgrad_lds_kernel(const TI* __restrict__ in, const TD* __restrict__ dout,
                                                                     float* __restrict__ part, int H, int W, int C, int RY) {
    extern __shared__ __attribute__((aligned(16))) float4 dw_smem[];
    const int WP = (W + 6) | 1, cq = C >> 2;  // slot pitch in pixels, odd: the tap rows of a 16-lane read group then sit in both
    float4* ring = dw_smem;                   // 128-byte halves of the bank period.  [8][WP][8]
    float4* dyb = dw_smem + 8 * WP * 8;       // [2][W][8]
    const int tid = threadIdx.x, lane = tid & 63, seg = __builtin_amdgcn_readfirstlane(tid >> 6), c4l = lane & 7, ti = lane >> 3;
    const int cc = blockIdx.x, yc = blockIdx.y, b = blockIdx.z, ny = gridDim.y;
    const int y0 = yc * RY, y1 = min(H, y0 + RY);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const long img = (long)b * H * W * cq + cc * 8;  // float4 index of (b, 0, 0, first quad of this chunk)
    constexpr int NT = XS * 64;
    // zero the pad columns of every slot once
    for (int q = tid; q < 8 * 6 * 8; q += NT) {
        const int slot = q / 48, r = q % 48, col = r / 8, c = r % 8;
        ring[(slot * WP + (col < 3 ? col : W + col)) * 8 + c] = zero;
    }
    const int items = W * 8;                       // float4 of one row of this channel chunk
    constexpr int MAXI = 2;                        // staged items per thread and tensor (the launcher checks W * 8 <= MAXI * NT)
    float4 rin[MAXI], rdy[MAXI];
    auto load_row = [&](int r, float4 (&v)[MAXI]) {  // input row r -> registers (zeros outside the image)
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int q = tid + k * NT;
            v[k] = (q < items && r >= 0 && r < H) ? ld4(in, img + ((long)r * W + q / 8) * cq + (q & 7)) : zero;
        }
    };
    auto store_row = [&](int r, const float4 (&v)[MAXI]) {
        float4* S = ring + ((r + 3) & 7) * WP * 8;
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int q = tid + k * NT;
            if (q < items) S[(3 + q / 8) * 8 + (q & 7)] = v[k];
        }
    };
    auto load_in = [&](int r) { load_row(r, rin); };
    auto store_in = [&](int r) { store_row(r, rin); };
    auto load_dy = [&](int y) {
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int q = tid + k * NT;
            rdy[k] = (q < items && y < y1) ? ld4(dout, img + ((long)y * W + q / 8) * cq + (q & 7)) : zero;
        }
    };
    auto store_dy = [&](int y) {
        float4* S = dyb + (y & 1) * W * 8;
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int q = tid + k * NT;
            if (q < items) S[q] = rdy[k];
        }
    };
    // prologue: rows y0 - 3 .. y0 + 2 into the ring; row y0 + 3 and dy row y0 wait in registers for the first barrier
    {
        float4 pre[6][MAXI];  // (all six rows in flight together: one memory round trip, not six)
#pragma unroll
        for (int k = 0; k < 6; ++k) load_row(y0 - 3 + k, pre[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) store_row(y0 - 3 + k, pre[k]);
    }
    load_in(y0 + 3);
    load_dy(y0);
    float4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = zero;
    const int xw = (W + XS - 1) / XS, xs0 = seg * xw, xs1 = min(W, xs0 + xw);
    for (int y = y0; y < y1; ++y) {
        store_in(y + 3);   // slot of row y - 5: nobody reads it any more (a slow thread is at most in row y - 1: rows y - 4 .. y + 2)
        store_dy(y);
        __syncthreads();
        load_in(y + 4);    // in flight under this row's arithmetic
        load_dy(y + 1);
        if (ti < 7) {
            const float4* S = ring + ((y + ti) & 7) * WP * 8 + c4l;   // input row y + ti - 3, padded column index = x + j
            const float4* D = dyb + (y & 1) * W * 8 + c4l;
            // seven positions per trip (the segments are 7 or 14 wide at every ConvNeXt stage): a 13-wide window, six register moves per
            // seven positions instead of six per position; a ragged tail multiplies zeros (dy) with clamped (in-row) window reads
            float4 win[13];
#pragma unroll
            for (int j = 0; j < 6; ++j) win[j] = S[(xs0 + j) * 8];
            for (int x = xs0; x < xs1; x += 7) {
                float4 d[7];
#pragma unroll
                for (int p = 0; p < 7; ++p) {
                    win[6 + p] = S[min(x + p + 6, W + 5) * 8];   // (a ragged tail re-reads the last, zeroed, pad column)
                    const float4 dv = D[min(x + p, xs1 - 1) * 8];   // (always an LDS read; the select is on the value)
                    d[p] = (x + p < xs1) ? dv : zero;
                }
#pragma unroll
                for (int p = 0; p < 7; ++p)
#pragma unroll
                    for (int j = 0; j < 7; ++j) fma4(acc[j], d[p], win[p + j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) win[j] = win[7 + j];
            }
        }
    }
    // add the x segments (fixed order) and store the chunk's 49 x 32 partial sums
    __syncthreads();
    float4* R = dw_smem;  // [XS][7][64] (the ring is free)
#pragma unroll
    for (int j = 0; j < 7; ++j) R[(seg * 7 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (seg == 0 && ti < 7) {
        float4* p = reinterpret_cast<float4*>(part + ((long)(b * ny + yc) * 49 + ti * 7) * C) + cc * 8 + c4l;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            float4 v = acc[j];
#pragma unroll
            for (int s2 = 1; s2 < XS; ++s2) {
                const float4 o = R[(s2 * 7 + j) * 64 + lane];
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            p[(long)j * cq] = v;
        }
    }
}

// ---- patchify -------------------------------------------------------------------------------------------------
// NCHW image -> rows [(b,py,px)][(c,ky,kx)]  (matches conv weight [O][c][ky][kx] flattened); V floats per access.
template <int V>
__global__ void patchify_nchw_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Cin, int H, int W,
                                     int k) {
    const int ph = (H + k - 1) / k, pw = (W + k - 1) / k, kv = k / V, K = Cin * k * k;
    const long total = (long)B * ph * pw * Cin * k * kv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int kx = (int)(r % kv) * V; r /= kv;
        const int ky = (int)(r % k); r /= k;
        const int c = (int)(r % Cin); r /= Cin;
        const int px = (int)(r % pw); r /= pw;
        const int py = (int)(r % ph);
        const int b = (int)(r / ph);
        const float* src = img + (((long)b * Cin + c) * H + py * k + ky) * W + px * k + kx;
        float* dst = out + ((long)(b * ph + py) * pw + px) * K + (c * k + ky) * k + kx;
        const bool in = py * k + ky < H && px * k + kx < W;  // W % V == 0: a vector is wholly inside or wholly in the zero pad
        if (V == 4) *reinterpret_cast<float4*>(dst) = in ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        else if (V == 2) *reinterpret_cast<float2*>(dst) = in ? *reinterpret_cast<const float2*>(src) : make_float2(0.f, 0.f);
        else *dst = in ? *src : 0.f;
    }
}

// NHWC feature <-> rows [(b,py,px)][(ky,kx,c)] for a k x k stride-k conv (weights permuted to [O][ky][kx][c]).
template <bool INVERSE>
__global__ void patchify_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C,
                                     int k) {
    const int ph = H / k, pw = W / k, cq = C >> 2;
    const long total = (long)B * H * W * cq;  // H, W multiples of k: every pixel belongs to exactly one patch
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int c4 = (int)(r % cq); r /= cq;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const long feat = (((long)b * H + y) * W + x) * cq + c4;
        const long prow = ((long)b * ph + y / k) * pw + x / k;
        const long pat = (prow * k * k + (y % k) * k + (x % k)) * cq + c4;
        if (INVERSE) reinterpret_cast<float4*>(dst)[feat] = reinterpret_cast<const float4*>(src)[pat];
        else reinterpret_cast<float4*>(dst)[pat] = reinterpret_cast<const float4*>(src)[feat];
    }
}

// ---- bilinear resize (align_corners = False) -------------------------------------------------------------------
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& lam) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}

__global__ void resize_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Hi, int Wi, int Ho,
                                  int Wo, int C) {
    const int cq = C >> 2;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const long total = (long)B * Ho * Wo * cq;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int c4 = (int)(r % cq); r /= cq;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, sy, Hi, y0, y1, ly);
        src_index(ox, sx, Wi, x0, x1, lx);
        const long base = (long)b * Hi * Wi;
        const float4 a = in4[(base + (long)y0 * Wi + x0) * cq + c4], bq = in4[(base + (long)y0 * Wi + x1) * cq + c4];
        const float4 c = in4[(base + (long)y1 * Wi + x0) * cq + c4], d = in4[(base + (long)y1 * Wi + x1) * cq + c4];
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        float4 o;
        o.x = w00 * a.x + w01 * bq.x + w10 * c.x + w11 * d.x;
        o.y = w00 * a.y + w01 * bq.y + w10 * c.y + w11 * d.y;
        o.z = w00 * a.z + w01 * bq.z + w10 * c.z + w11 * d.z;
        o.w = w00 * a.w + w01 * bq.w + w10 * c.w + w11 * d.w;
        reinterpret_cast<float4*>(out)[idx] = o;
    }
}

// gather form of the adjoint: every input pixel sums the output pixels whose 2x2 footprint touches it
__global__ void resize_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int Hi, int Wi, int Ho,
                                  int Wo, int C) {
    const int cq = C >> 2;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const long total = (long)B * Hi * Wi * cq;
    const float4* d4 = reinterpret_cast<const float4*>(dout);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int c4 = (int)(r % cq); r /= cq;
        const int ix = (int)(r % Wi); r /= Wi;
        const int iy = (int)(r % Hi);
        const int b = (int)(r / Hi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = 0; oy < Ho; ++oy) {
            int y0, y1; float ly;
            src_index(oy, sy, Hi, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 0; ox < Wo; ++ox) {
                int x0, x1; float lx;
                src_index(ox, sx, Wi, x0, x1, lx);
                const float w = wy * ((x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f));
                if (w == 0.f) continue;
                const float4 g = d4[(((long)b * Ho + oy) * Wo + ox) * cq + c4];
                acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
            }
        }
        reinterpret_cast<float4*>(din)[idx] = acc;
    }
}

__global__ void set_rows_kernel(float* __restrict__ out, const float* __restrict__ vec, int B, long row_stride, int D) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[(i / D) * row_stride + i % D] = vec[i % D];
}

inline int blocks_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
}  // namespace

template <typename TI, typename TA, typename TO>
static int launch_dwconv7(const TI* in, const float* w_tap_major, const float* bias, const TA* add, TO* out, int B, int H, int W,
                          int C, int flip, void* stream) {
    if (!in || !w_tap_major || !out || B <= 0 || H <= 0 || W <= 0) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_UNSUPPORTED;
    const long total = (long)B * H * ((W + 6) / 7) * (C / 4);
    // 32-bit float4 / thread indexing inside the kernel
    if ((long)B * H * W * (C / 4) >= (1L << 32) || total >= (1L << 31)) return DKD_ERR_UNSUPPORTED;
    static const int force = getenv("DKD_DWCONV_RY") ? atoi(getenv("DKD_DWCONV_RY")) : -1;  // A/B: 0 = first form
    const int cq = C / 4;
    int CB = 0;  // largest divisor of cq that is <= 64: the block's channel-quad range
    for (int d = 1; d <= 64 && d <= cq; ++d)
        if (cq % d == 0) CB = d;
    // measured (scripts/bench_dwconv.py, bs 64): RY = 3 wins at 56 x 56 and 28 x 28 (fwd 77 -> 53 us, 42 -> 30 us) and for the
    // forward at 14 x 14; at 7 x 7 (and the dgrad at 14 x 14) there are too few tiles for the bigger per-thread footprint.
    // 4-wide strips (more threads, less reuse) and RY = 4 (one wave per SIMD) were slower.
    const int ry = force >= 0 ? force : (H >= 28 || (H >= 14 && !flip) ? 3 : 0), xw = 7;
    if (ry >= 2 && CB >= 16) {
        const int SP = 256 / CB, chunks = cq / CB;
        const int nyb = (H + ry - 1) / ry;
        const long tiles = (long)B * nyb * ((W + xw - 1) / xw);
        long gx = (tiles + SP - 1) / SP;
        const long cap = 3072 / chunks > 1 ? 3072 / chunks : 1;  // ~12 blocks per CU in flight over all chunks, then stride
        if (gx > cap) gx = cap;
        const dim3 grid((unsigned)gx, chunks), block(256);
        const size_t lds = (size_t)49 * CB * sizeof(float4);
        hipStream_t st = (hipStream_t)stream;
#define DKD_DW(FL, R, X) hipLaunchKernelGGL((dwconv7_lds_kernel<FL, R, X, TI, TA, TO>), grid, block, lds, st, in, w_tap_major, bias, add, out, B, H, W, C, CB, SP)
#define DKD_DW2(FL, R) DKD_DW(FL, R, 7)
        if (flip) { if (ry >= 4) { DKD_DW2(true, 4); } else if (ry == 3) { DKD_DW2(true, 3); } else { DKD_DW2(true, 2); } }
        else { if (ry >= 4) { DKD_DW2(false, 4); } else if (ry == 3) { DKD_DW2(false, 3); } else { DKD_DW2(false, 2); } }
#undef DKD_DW2
#undef DKD_DW
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
    if (flip)
        hipLaunchKernelGGL((dwconv7_kernel<true, TI, TA, TO>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           w_tap_major, bias, add, out, B, H, W, C);
    else
        hipLaunchKernelGGL((dwconv7_kernel<false, TI, TA, TO>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in,
                           w_tap_major, bias, add, out, B, H, W, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dwconv7_fwd_f32(const float* in, const float* w_tap_major, const float* bias, const float* add,
                                   float* out, int B, int H, int W, int C, int flip, void* stream) {
    return launch_dwconv7(in, w_tap_major, bias, add, out, B, H, W, C, flip, stream);
}

// mixed-precision step: bf16 input; addend and output fp32 or bf16 (add16 / out16); taps and bias fp32 (pre-rounded by the caller)
extern "C" int dkd_dwconv7_fwd_io(const void* in_bf16, const float* w_tap_major, const float* bias, const void* add, int add16,
                                  void* out, int out16, int B, int H, int W, int C, int flip, void* stream) {
    const __bf16* in = reinterpret_cast<const __bf16*>(in_bf16);
    int rc = DKD_ERR_BAD_ARG;
    DKD_IO_TYPE(add16, TA, DKD_IO_TYPE(out16, TO, rc = launch_dwconv7(in, w_tap_major, bias, reinterpret_cast<const TA*>(add),
                                                                          reinterpret_cast<TO*>(out), B, H, W, C, flip, stream)));
    return rc;
}

// row chunks per image: two for the 56 x 56 stage (its 32-channel chunks alone leave the chip a third empty), one otherwise; a chunk
// re-reads six halo rows of x
static int dw_ry(int H) { return H >= 56 ? (H + 1) / 2 : H; }
extern "C" long dkd_dwconv7_wgrad_f32_parts(int B, int H) { return (long)B * ((H + dw_ry(H) - 1) / dw_ry(H)); }

template <typename TI, typename TD>
static int launch_dwconv7_wgrad(const TI* in, const TD* dout, float* partials, int B, int H, int W, int C, void* stream) {
    if (!in || !dout || !partials || B <= 0 || H <= 0 || W <= 0) return DKD_ERR_BAD_ARG;
    const int RY = dw_ry(H), ny = (H + RY - 1) / RY;
    if (C & 3) return DKD_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    static const bool old_kernel = getenv("DKD_DWCONV7_WGRAD_OLD") != nullptr;  // A/B: the one-thread-per-tap-row kernel
    const int XS = W >= 28 ? 4 : (W >= 14 ? 2 : 1);
    size_t lds = ((size_t)8 * ((W + 6) | 1) * 8 + (size_t)2 * W * 8) * sizeof(float4);   // ring + dy double buffer ...
    if (lds < (size_t)XS * 7 * 64 * sizeof(float4)) lds = (size_t)XS * 7 * 64 * sizeof(float4);  // ... re-used by the segment sums
    // whole 32-channel chunks, a row of the chunk within two staged float4 per thread, the ring within the CU's LDS
    if (!old_kernel && C % DW_CC == 0 && W * 8 <= 2 * XS * 64 && lds <= 150 * 1024 && B <= 65535) {
        const dim3 grid(C / DW_CC, ny, B);
#define DKD_DW_WG(XSV)                                                                                                             \
    do {                                                                                                                           \
        auto kern = dwconv7_wgrad_lds_kernel<TI, TD, XSV>;                                                                         \
        if (lds > 64 * 1024) {                                                                                                     \
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return (int)e;                                                                                    \
        }                                                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(XSV * 64), lds, st, in, dout, partials, H, W, C, RY);                                  \
    } while (0)
        if (XS == 4) DKD_DW_WG(4);
        else if (XS == 2) DKD_DW_WG(2);
        else DKD_DW_WG(1);
#undef DKD_DW_WG
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
    const long total = (long)B * ny * 7 * (C / 4);
    hipLaunchKernelGGL((dwconv7_wgrad_kernel<TI, TD>), dim3(blocks_for(total)), dim3(256), 0, st, in, dout, partials, B, H, W, C, RY);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dwconv7_wgrad_f32(const float* in, const float* dout, float* partials, int B, int H, int W, int C,
                                     void* stream) {
    return launch_dwconv7_wgrad(in, dout, partials, B, H, W, C, stream);
}

extern "C" int dkd_dwconv7_wgrad_bf16(const void* in, const void* dout, float* partials, int B, int H, int W, int C, void* stream) {
    return launch_dwconv7_wgrad(reinterpret_cast<const __bf16*>(in), reinterpret_cast<const __bf16*>(dout), partials, B, H, W, C, stream);
}

extern "C" int dkd_patchify_nchw_f32(const float* img, float* out, int B, int Cin, int H, int W, int k, void* stream) {
    if (!img || !out || B <= 0 || k <= 0 || H <= 0 || W <= 0) return DKD_ERR_BAD_ARG;
    const int V = (k % 4 == 0 && W % 4 == 0) ? 4 : ((k % 2 == 0 && W % 2 == 0) ? 2 : 1);
    const long total = (long)B * Cin * ((H + k - 1) / k * k) * ((W + k - 1) / k * k) / V;
    dim3 grid(blocks_for(total)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (V == 4) hipLaunchKernelGGL(patchify_nchw_kernel<4>, grid, block, 0, st, img, out, B, Cin, H, W, k);
    else if (V == 2) hipLaunchKernelGGL(patchify_nchw_kernel<2>, grid, block, 0, st, img, out, B, Cin, H, W, k);
    else hipLaunchKernelGGL(patchify_nchw_kernel<1>, grid, block, 0, st, img, out, B, Cin, H, W, k);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_patchify_nhwc_f32(const float* src, float* dst, int B, int H, int W, int C, int k, int inverse,
                                     void* stream) {
    if (!src || !dst || B <= 0 || k <= 0 || H % k || W % k) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_ALIGN;
    const long total = (long)B * H * W * (C / 4);
    if (inverse)
        hipLaunchKernelGGL(patchify_nhwc_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst,
                           B, H, W, C, k);
    else
        hipLaunchKernelGGL(patchify_nhwc_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, src,
                           dst, B, H, W, C, k);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_resize_bilinear_fwd_f32(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                           void* stream) {
    if (!in || !out || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_ALIGN;
    const long total = (long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(resize_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, B, Hi, Wi,
                       Ho, Wo, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_resize_bilinear_bwd_f32(const float* dout, float* din, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                           void* stream) {
    if (!dout || !din || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return DKD_ERR_BAD_ARG;
    if (C & 3) return DKD_ERR_ALIGN;
    const long total = (long)B * Hi * Wi * (C / 4);
    hipLaunchKernelGGL(resize_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dout, din, B, Hi, Wi,
                       Ho, Wo, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_set_rows_f32(float* out, const float* vec, int B, long row_stride, int D, void* stream) {
    if (!out || !vec || B <= 0 || D <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(set_rows_kernel, dim3(blocks_for((long)B * D)), dim3(256), 0, (hipStream_t)stream, out, vec, B,
                       row_stride, D);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
