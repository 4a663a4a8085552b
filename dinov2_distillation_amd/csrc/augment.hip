// GPU input pipeline: the reference's DataAugmentationDINO (datasets/augmentations.py:23-78 - RandomResizedCrop
// bicubic, horizontal flip, RandAugment 9 x magnitude 9, ToTensor, Normalize, RandomErasing) on decoded uint8 images, a whole
// batch per launch.  The reference runs it in 8 PIL worker processes; at ~1450 images/s per GPU that pipeline cannot feed one
// MI355X.  Every pixel operation reproduces Pillow's arithmetic BIT FOR BIT (the CPU oracle calls Pillow itself):
//   * bicubic resize: Pillow's two-pass fixed-point resample (22-bit integer coefficients, horizontal pass rounded to uint8,
//     then the vertical pass); the coefficient tables are computed on the host in double precision exactly as
//     libImaging/Resample.c does, the kernels are pure integer MACs;
//   * affine ops (shear / translate / rotate): libImaging/Geometry.c's double-precision inverse map + bilinear filter,
//     evaluated in fp64 with explicitly rounded operations (no FMA contraction), truncated to uint8;
//   * enhancers: Image.blend's float32 arithmetic (truncation inside [0,1], clipping outside), SMOOTH as the 3x3 float32
//     filter with Pillow's summation order, L = (19595 R + 38470 G + 7471 B + 32768) >> 16;
//   * posterize / solarize / autocontrast / equalize: 256-entry look-up tables built per image and channel from integer
//     histograms (integer atomics: order-independent, deterministic).
// Images travel as RGBX (4 bytes per pixel).  HBM-bound by construction: ~30 launches per batch, a few bytes per pixel each.
#include "common.h"
#include "dkd_hip.h"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8i(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- resize ----------------------------------------------------------------------------------------------------------------
// meta[b] = {src offset (bytes, low / high word), H, W, crop top, crop left, crop h, crop w, flip}
struct ImgMeta {
    long off;
    int H, W, ci, cj, ch, cw, flip;
};
__device__ __forceinline__ ImgMeta load_meta(const int* __restrict__ meta, int b) {
    const int* m = meta + b * 10;
    ImgMeta r;
    r.off = ((long)m[1] << 32) | (unsigned int)m[0];
    r.H = m[2]; r.W = m[3]; r.ci = m[4]; r.cj = m[5]; r.ch = m[6]; r.cw = m[7]; r.flip = m[8];
    return r;
}

// Pillow's coefficient tables on the device: libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic
// filter, one thread per (table, output position), the same double-precision operations in the same order (this file is
// compiled with -ffp-contract=off; IEEE add / mul / div are correctly rounded on both sides, so the integers are identical -
// tests/test_augment_gpu.py checks them against the host tables).  sizes[t] = input extent of table t (crop width / height).
__device__ __forceinline__ double bicubic_w(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
__global__ void resample_coeffs_kernel(const int* __restrict__ sizes, int* __restrict__ kk, int* __restrict__ bounds, int T, int S, int ksize) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * S) return;
    const int t = i / S, xx = i % S;
    const int in_size = sizes[t];
    const double scale = (double)in_size / (double)S;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double center = 0.0 + ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = kk + (long)i * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_w(((double)(x + xmin) - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            double w = bicubic_w(((double)(x + xmin) - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            v = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
        }
        k[x] = v;
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xmax;
}

// horizontal pass: tmp[b][y][xx] (y = crop row) = clip8((2^21 + sum_x src[y][xmin + x] * k[xx][x]) >> 22) per channel
__global__ void resample_h_kernel(const unsigned char* __restrict__ src, const int* __restrict__ meta, const int* __restrict__ kk,
                                  const int* __restrict__ bounds, uchar4* __restrict__ tmp, int B, int S, int ksize, int max_h) {
    const long total = (long)B * max_h * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % S);
        const int y = (int)((i / S) % max_h);
        const int b = (int)(i / ((long)S * max_h));
        const ImgMeta m = load_meta(meta, b);
        if (y >= m.ch) continue;
        const int xmin = bounds[(b * S + xx) * 2], xmax = bounds[(b * S + xx) * 2 + 1];
        const int* k = kk + ((long)b * S + xx) * ksize;
        const unsigned char* row = src + m.off + ((long)(m.ci + y) * m.W + m.cj + xmin) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < xmax; ++x) {
            const int w = k[x];
            s0 += row[3 * x] * w;
            s1 += row[3 * x + 1] * w;
            s2 += row[3 * x + 2] * w;
        }
        tmp[((long)b * max_h + y) * S + xx] = make_uchar4(clip8i(s0 >> PRECISION_BITS), clip8i(s1 >> PRECISION_BITS), clip8i(s2 >> PRECISION_BITS), 0);
    }
}

// vertical pass (+ horizontal flip of the result): img[b][yy][xx'] from tmp[b][ymin + y][xx]
__global__ void resample_v_kernel(const uchar4* __restrict__ tmp, const int* __restrict__ meta, const int* __restrict__ kk,
                                  const int* __restrict__ bounds, uchar4* __restrict__ img, int B, int S, int ksize, int max_h) {
    const long total = (long)B * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % S);
        const int yy = (int)((i / S) % S);
        const int b = (int)(i / ((long)S * S));
        const ImgMeta m = load_meta(meta, b);
        const int ymin = bounds[(b * S + yy) * 2], ymax = bounds[(b * S + yy) * 2 + 1];
        const int* k = kk + ((long)b * S + yy) * ksize;
        const uchar4* col = tmp + ((long)b * max_h + ymin) * S + xx;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < ymax; ++y) {
            const uchar4 p = col[(long)y * S];
            const int w = k[y];
            s0 += p.x * w; s1 += p.y * w; s2 += p.z * w;
        }
        const int xo = m.flip ? S - 1 - xx : xx;
        img[((long)b * S + yy) * S + xo] = make_uchar4(clip8i(s0 >> PRECISION_BITS), clip8i(s1 >> PRECISION_BITS), clip8i(s2 >> PRECISION_BITS), 0);
    }
}

// ---- RandAugment round --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lum(uchar4 p) { return (p.x * 19595 + p.y * 38470 + p.z * 7471 + 0x8000) >> 16; }

// per image: histograms of the three channels and the sum of L (only images whose op needs them take part)
__global__ __launch_bounds__(256) void aug_stats_kernel(const uchar4* __restrict__ img, const int* __restrict__ ops, int* __restrict__ hist,
                                                        unsigned long long* __restrict__ lsum, int S, int chunks) {
    __shared__ int h[3 * 256];
    __shared__ unsigned long long ls;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int op = ops[b];
    if (op != DKD_AUG_CONTRAST && op != DKD_AUG_AUTOCONTRAST && op != DKD_AUG_EQUALIZE) return;
    for (int i = threadIdx.x; i < 768; i += 256) h[i] = 0;
    if (threadIdx.x == 0) ls = 0ull;
    __syncthreads();
    const int n = S * S, per = (n + chunks - 1) / chunks;
    const int lo = chunk * per, hi = min(n, lo + per);
    unsigned long long mine = 0ull;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const uchar4 p = img[(long)b * n + i];
        atomicAdd(&h[p.x], 1);
        atomicAdd(&h[256 + p.y], 1);
        atomicAdd(&h[512 + p.z], 1);
        mine += (unsigned long long)lum(p);
    }
    atomicAdd(&ls, mine);
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256)
        if (h[i]) atomicAdd(&hist[b * 768 + i], h[i]);
    if (threadIdx.x == 0) atomicAdd(&lsum[b], ls);
}

// per image: the 3 x 256 look-up table of a LUT op, or the L mean of Contrast (ImageStat mean, int(mean + 0.5))
__global__ __launch_bounds__(256) void aug_lut_kernel(const int* __restrict__ ops, const double* __restrict__ fparams, const int* __restrict__ hist,
                                                      const unsigned long long* __restrict__ lsum, unsigned char* __restrict__ lut,
                                                      int* __restrict__ meanL, int S) {
    __shared__ int h[256];
    __shared__ long pre[257];
    __shared__ int lo_s, hi_s, nz_s, last_s;
    const int b = blockIdx.x, t = threadIdx.x, op = ops[b];
    const double p0 = fparams[b * 8];
    if (op == DKD_AUG_CONTRAST) {
        if (t == 0) {
            const double mean = __ddiv_rn((double)lsum[b], (double)((long)S * S));
            meanL[b] = (int)__dadd_rn(mean, 0.5);
        }
        return;
    }
    if (op == DKD_AUG_POSTERIZE) {
        const int bits = (int)p0;
        const unsigned char mask = (unsigned char)(~((1 << (8 - bits)) - 1));
        for (int c = 0; c < 3; ++c) lut[(b * 3 + c) * 256 + t] = (unsigned char)t & mask;
        return;
    }
    if (op == DKD_AUG_SOLARIZE) {
        const unsigned char v = (double)t < p0 ? (unsigned char)t : (unsigned char)(255 - t);
        for (int c = 0; c < 3; ++c) lut[(b * 3 + c) * 256 + t] = v;
        return;
    }
    if (op != DKD_AUG_AUTOCONTRAST && op != DKD_AUG_EQUALIZE) return;
    for (int c = 0; c < 3; ++c) {
        h[t] = hist[b * 768 + c * 256 + t];
        __syncthreads();
        if (t == 0) {  // 256 entries: a serial scan is a few hundred cycles
            int lo = -1, hi = -1, nz = 0, last = 0;
            long acc = 0;
            for (int i = 0; i < 256; ++i) {
                pre[i] = acc;
                acc += h[i];
                if (h[i]) { if (lo < 0) lo = i; hi = i; ++nz; last = h[i]; }
            }
            pre[256] = acc;
            lo_s = lo; hi_s = hi; nz_s = nz; last_s = last;
        }
        __syncthreads();
        int v = t;
        if (op == DKD_AUG_AUTOCONTRAST) {  // ImageOps.autocontrast(cutoff = 0): lut = int(ix * scale + offset), clipped
            if (hi_s > lo_s) {
                const double scale = __ddiv_rn(255.0, (double)(hi_s - lo_s));
                const double offset = __dmul_rn(-(double)lo_s, scale);
                const int ix = (int)__dadd_rn(__dmul_rn((double)t, scale), offset);
                v = ix < 0 ? 0 : (ix > 255 ? 255 : ix);
            }
        } else {  // ImageOps.equalize: step = (sum(histo) - histo[-1]) // 255; lut[i] = (step // 2 + sum(h[:i])) // step
            if (nz_s > 1) {
                const long step = (pre[256] - last_s) / 255;
                if (step) v = (int)((step / 2 + pre[t]) / step);
            }
            v = v > 255 ? 255 : v;
        }
        lut[(b * 3 + c) * 256 + t] = (unsigned char)v;
        __syncthreads();
    }
}

// Image.blend(degenerate, image, alpha) of libImaging/Blend.c on one channel: float32, truncation / clipping as Pillow does
__device__ __forceinline__ unsigned char blend1(int d, int s, float alpha, bool inside) {
    const float t = __fadd_rn((float)d, __fmul_rn(alpha, (float)(s - d)));
    if (inside) return (unsigned char)(int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)(int)t);
}

// ImageFilter.SMOOTH at an interior pixel: (1 1 1 / 1 5 1 / 1 1 1) / 13 in float32, rows bottom-up, + 0.5, clip8
__device__ __forceinline__ float k1x3(float a, float b, float c, float k0, float k1, float k2) {
    return __fadd_rn(__fadd_rn(__fmul_rn(a, k0), __fmul_rn(b, k1)), __fmul_rn(c, k2));
}
__device__ __forceinline__ unsigned char clip8f(float v) { return v <= 0.f ? 0 : (v >= 255.f ? 255 : (unsigned char)(int)v); }

__device__ __forceinline__ unsigned char chan(uchar4 p, int c) { return c == 0 ? p.x : (c == 1 ? p.y : p.z); }

// bilinear_filter32RGB of libImaging/Geometry.c at (xin, yin), one channel, in fp64 without contraction
__device__ __forceinline__ unsigned char bilinear1(const uchar4* __restrict__ im, int S, int x, int y, double dx, double dy, int c) {
    const int x0 = x < 0 ? 0 : (x >= S ? S - 1 : x), x1 = x + 1 < 0 ? 0 : (x + 1 >= S ? S - 1 : x + 1);
    const int yc = y < 0 ? 0 : (y >= S ? S - 1 : y);
    const double a = (double)chan(im[yc * S + x0], c), b = (double)chan(im[yc * S + x1], c);
    double v1 = __dadd_rn(a, __dmul_rn(__dsub_rn(b, a), dx));
    double v2 = v1;
    if (y + 1 >= 0 && y + 1 < S) {
        const double a2 = (double)chan(im[(y + 1) * S + x0], c), b2 = (double)chan(im[(y + 1) * S + x1], c);
        v2 = __dadd_rn(a2, __dmul_rn(__dsub_rn(b2, a2), dx));
    }
    return (unsigned char)(int)__dadd_rn(v1, __dmul_rn(__dsub_rn(v2, v1), dy));
}

__global__ __launch_bounds__(256) void aug_apply_kernel(const uchar4* __restrict__ in, uchar4* __restrict__ out, const int* __restrict__ ops,
                                                        const double* __restrict__ fparams, const unsigned char* __restrict__ lut,
                                                        const int* __restrict__ meanL, int B, int S) {
    const long total = (long)B * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S), b = (int)(i / ((long)S * S));
        const int op = ops[b];
        const uchar4* im = in + (long)b * S * S;
        const uchar4 p = im[y * S + x];
        uchar4 o = p;
        const double* fp = fparams + b * 8;
        if (op == DKD_AUG_AFFINE) {  // affine_transform + bilinear_filter32RGB; outside the source: 0
            const double xc = __dadd_rn((double)x, 0.5), yc = __dadd_rn((double)y, 0.5);
            const double xin = __dadd_rn(__dadd_rn(__dmul_rn(fp[0], xc), __dmul_rn(fp[1], yc)), fp[2]);
            const double yin = __dadd_rn(__dadd_rn(__dmul_rn(fp[3], xc), __dmul_rn(fp[4], yc)), fp[5]);
            if (xin < 0.0 || xin >= (double)S || yin < 0.0 || yin >= (double)S) {
                o = make_uchar4(0, 0, 0, 0);
            } else {
                const double xs = __dsub_rn(xin, 0.5), ys = __dsub_rn(yin, 0.5);
                const double fx = floor(xs), fy = floor(ys);
                const int xi = (int)fx, yi = (int)fy;
                const double dx = __dsub_rn(xs, fx), dy = __dsub_rn(ys, fy);
                o = make_uchar4(bilinear1(im, S, xi, yi, dx, dy, 0), bilinear1(im, S, xi, yi, dx, dy, 1), bilinear1(im, S, xi, yi, dx, dy, 2), 0);
            }
        } else if (op == DKD_AUG_POSTERIZE || op == DKD_AUG_SOLARIZE || op == DKD_AUG_AUTOCONTRAST || op == DKD_AUG_EQUALIZE) {
            const unsigned char* l = lut + b * 768;
            o = make_uchar4(l[p.x], l[256 + p.y], l[512 + p.z], 0);
        } else if (op == DKD_AUG_BRIGHTNESS || op == DKD_AUG_COLOR || op == DKD_AUG_CONTRAST || op == DKD_AUG_SHARPNESS) {
            const float alpha = (float)fp[0];  // ImagingBlend takes a C float
            const bool inside = alpha >= 0.f && alpha <= 1.f;
            int d0 = 0, d1 = 0, d2 = 0;
            if (op == DKD_AUG_COLOR) d0 = d1 = d2 = lum(p);
            else if (op == DKD_AUG_CONTRAST) d0 = d1 = d2 = meanL[b];
            else if (op == DKD_AUG_SHARPNESS) {
                d0 = p.x; d1 = p.y; d2 = p.z;
                if (x > 0 && x < S - 1 && y > 0 && y < S - 1) {
                    const float k1 = __fdiv_rn(1.f, 13.f), k5 = __fdiv_rn(5.f, 13.f);
                    const uchar4 r0[3] = {im[(y + 1) * S + x - 1], im[(y + 1) * S + x], im[(y + 1) * S + x + 1]};
                    const uchar4 r1[3] = {im[y * S + x - 1], p, im[y * S + x + 1]};
                    const uchar4 r2[3] = {im[(y - 1) * S + x - 1], im[(y - 1) * S + x], im[(y - 1) * S + x + 1]};
                    int d[3];
                    for (int c = 0; c < 3; ++c) {
                        float ss = 0.5f;
                        ss = __fadd_rn(ss, k1x3((float)chan(r0[0], c), (float)chan(r0[1], c), (float)chan(r0[2], c), k1, k1, k1));
                        ss = __fadd_rn(ss, k1x3((float)chan(r1[0], c), (float)chan(r1[1], c), (float)chan(r1[2], c), k1, k5, k1));
                        ss = __fadd_rn(ss, k1x3((float)chan(r2[0], c), (float)chan(r2[1], c), (float)chan(r2[2], c), k1, k1, k1));
                        d[c] = clip8f(ss);
                    }
                    d0 = d[0]; d1 = d[1]; d2 = d[2];
                }
            }
            o = make_uchar4(blend1(d0, p.x, alpha, inside), blend1(d1, p.y, alpha, inside), blend1(d2, p.z, alpha, inside), 0);
        }
        out[i] = o;
    }
}

// ToTensor (/255) + Normalize ((v - mean) / std, float32, the two rounded operations torch performs) + RandomErasing (value 0)
__global__ __launch_bounds__(256) void aug_finish_kernel(const uchar4* __restrict__ img, const int* __restrict__ erase, float* __restrict__ out,
                                                         int B, int S, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long total = (long)B * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S), b = (int)(i / ((long)S * S));
        const uchar4 p = img[i];
        const int* e = erase + b * 4;  // top, left, h, w (h == 0: no erasing)
        const bool er = e[2] > 0 && y >= e[0] && y < e[0] + e[2] && x >= e[1] && x < e[1] + e[3];
        const long plane = (long)S * S, base = (long)b * 3 * plane + (long)y * S + x;
        out[base] = er ? 0.f : __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.x, 255.f), m0), s0);
        out[base + plane] = er ? 0.f : __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.y, 255.f), m1), s1);
        out[base + 2 * plane] = er ? 0.f : __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.z, 255.f), m2), s2);
    }
}

int grid_1d(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 16384 ? (b < 1 ? 1 : b) : 16384);
}
}  // namespace

extern "C" int dkd_aug_resize(const void* src, const int* meta, const int* kh, const int* bh, const int* kv, const int* bv, void* tmp,
                              void* img, int B, int S, int ksize_h, int ksize_v, int max_h, void* stream) {
    if (!src || !meta || !kh || !bh || !kv || !bv || !tmp || !img || B <= 0 || S <= 0 || ksize_h <= 0 || ksize_v <= 0 || max_h <= 0)
        return DKD_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_1d((long)B * max_h * S)), dim3(256), 0, st, (const unsigned char*)src, meta, kh, bh,
                       (uchar4*)tmp, B, S, ksize_h, max_h);
    DKD_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_1d((long)B * S * S)), dim3(256), 0, st, (const uchar4*)tmp, meta, kv, bv, (uchar4*)img, B,
                       S, ksize_v, max_h);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_aug_coeffs(const int* sizes, int* kk, int* bounds, int T, int S, int ksize, void* stream) {
    if (!sizes || !kk || !bounds || T <= 0 || S <= 0 || ksize <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3((T * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, sizes, kk, bounds, T, S, ksize);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_aug_round(const void* in, void* out, const int* ops, const double* fparams, int* hist, void* lsum, void* lut, int* meanL,
                             int B, int S, int need_stats, void* stream) {
    if (!in || !out || !ops || !fparams || !hist || !lsum || !lut || !meanL || B <= 0 || S <= 0) return DKD_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (need_stats) {
        hipError_t e = hipMemsetAsync(hist, 0, (size_t)B * 768 * sizeof(int), st);
        if (e == hipSuccess) e = hipMemsetAsync(lsum, 0, (size_t)B * sizeof(unsigned long long), st);
        if (e != hipSuccess) return (int)e;
        const int chunks = 8;
        hipLaunchKernelGGL(aug_stats_kernel, dim3(B * chunks), dim3(256), 0, st, (const uchar4*)in, ops, hist, (unsigned long long*)lsum, S, chunks);
        DKD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(aug_lut_kernel, dim3(B), dim3(256), 0, st, ops, fparams, hist, (const unsigned long long*)lsum, (unsigned char*)lut, meanL, S);
    DKD_LAUNCH_CHECK();
    hipLaunchKernelGGL(aug_apply_kernel, dim3(grid_1d((long)B * S * S)), dim3(256), 0, st, (const uchar4*)in, (uchar4*)out, ops, fparams,
                       (const unsigned char*)lut, meanL, B, S);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_aug_finish(const void* img, const int* erase, float* out, int B, int S, float mean0, float mean1, float mean2, float std0,
                              float std1, float std2, void* stream) {
    if (!img || !erase || !out || B <= 0 || S <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(aug_finish_kernel, dim3(grid_1d((long)B * S * S)), dim3(256), 0, (hipStream_t)stream, (const uchar4*)img, erase, out, B,
                       S, mean0, mean1, mean2, std0, std1, std2);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
