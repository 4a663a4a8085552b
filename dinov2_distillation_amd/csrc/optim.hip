// Optimiser tail of the step: global gradient L2 norm, clip-by-norm and AdamW fused over FLAT fp32 buffers
// (all trainable parameters of the student and of the ScaleKD projectors live in one contiguous HBM arena, their
// gradients in a second one: one launch each, 16 B per lane, no host synchronisation - the clip coefficient is read
// from device memory).  Replaces torch.optim.AdamW (train/distillation_module.py:476-479) and Lightning's
// gradient_clip_val=1.0 / algorithm "norm" (train.py:267-268).
#include "common.h"
#include "dkd_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float4* __restrict__ x, long nvec, float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = (float)t;         // sum of squares
        out[1] = (float)sqrt(t);   // L2 norm
    }
}

// p16 (nullable): a bf16 shadow of the parameter arena, written with the update - the 16-bit operand copies torch autocast makes
// of every weight at every use (the mixed-precision step, functional16.py) cost no launch and no extra read of the arena
__global__ __launch_bounds__(256) void adamw_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                    float4* __restrict__ m, float4* __restrict__ v, long nvec, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, const float* __restrict__ gnorm, float max_norm,
                                                    dkd_bf16x4* __restrict__ p16) {
    float clip = 1.f;
    if (gnorm) clip = fminf(1.f, max_norm / (gnorm[1] + 1e-6f));
    const float decay = 1.f - lr * wd, step = lr * inv_bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define DKD_ADAMW_LANE(c)                                          \
    {                                                              \
        const float gr = gg.c * clip;                              \
        mm.c = b1 * mm.c + (1.f - b1) * gr;                        \
        vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                   \
        const float den = sqrtf(vv.c) * inv_sqrt_bc2 + eps;        \
        pp.c = pp.c * decay - step * (mm.c / den);                 \
    }
        DKD_ADAMW_LANE(x) DKD_ADAMW_LANE(y) DKD_ADAMW_LANE(z) DKD_ADAMW_LANE(w)
#undef DKD_ADAMW_LANE
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (p16) p16[i] = dkd_bf16x4{(__bf16)pp.x, (__bf16)pp.y, (__bf16)pp.z, (__bf16)pp.w};
    }
}

// Same update with every scalar read from DEVICE memory, hyper = {lr, beta1, beta2, eps, weight_decay, step}: nothing about
// the optimiser state is baked into the launch arguments, so a captured hipGraph of the training step stays valid while the
// step counter advances and a scheduler changes the learning rate (both live in `hyper`).
__global__ __launch_bounds__(256) void adamw_dev_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                        float4* __restrict__ m, float4* __restrict__ v, long nvec,
                                                        const float* __restrict__ hyper, const float* __restrict__ gnorm,
                                                        float max_norm, dkd_bf16x4* __restrict__ p16) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], t = hyper[5];
    const float inv_bc1 = 1.f / (1.f - powf(b1, t)), inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, t));
    float clip = 1.f;
    if (gnorm) clip = fminf(1.f, max_norm / (gnorm[1] + 1e-6f));
    const float decay = 1.f - lr * wd, step = lr * inv_bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define DKD_ADAMW_LANE(c)                                          \
    {                                                              \
        const float gr = gg.c * clip;                              \
        mm.c = b1 * mm.c + (1.f - b1) * gr;                        \
        vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                   \
        const float den = sqrtf(vv.c) * inv_sqrt_bc2 + eps;        \
        pp.c = pp.c * decay - step * (mm.c / den);                 \
    }
        DKD_ADAMW_LANE(x) DKD_ADAMW_LANE(y) DKD_ADAMW_LANE(z) DKD_ADAMW_LANE(w)
#undef DKD_ADAMW_LANE
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (p16) p16[i] = dkd_bf16x4{(__bf16)pp.x, (__bf16)pp.y, (__bf16)pp.z, (__bf16)pp.w};
    }
}

__global__ void scale_by_clip_kernel(float4* __restrict__ g, long nvec, const float* __restrict__ gnorm, float max_norm) {
    const float clip = fminf(1.f, max_norm / (gnorm[1] + 1e-6f));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        float4 v = g[i];
        v.x *= clip; v.y *= clip; v.z *= clip; v.w *= clip;
        g[i] = v;
    }
}

// Small derived copies of parameters that the mixed-precision step needs once per optimiser step (depthwise taps transposed to
// tap-major and rounded, rounded biases, patch-conv weights re-ordered and cast): ONE launch for all of them instead of two or
// three per parameter.  table[e] = {src (float*), dst, batch, rows, cols, flags, 0, 0} as int64: dst[b][c][r] = f(src[b][r][c]),
// flags 1: round to the nearest bf16 (fp32 dst holds bf16 values), 2: dst is bf16.
__global__ __launch_bounds__(256) void prep_batch_kernel(const long* __restrict__ table) {
    const long* e = table + 8 * (long)blockIdx.y;
    const float* __restrict__ src = reinterpret_cast<const float*>(e[0]);
    const long batch = e[2], rows = e[3], cols = e[4], flags = e[5], per = rows * cols, total = batch * per;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long b = o / per, rem = o - b * per, c = rem / rows, r = rem - c * rows;
        const float x = src[b * per + r * cols + c];
        if (flags & 2) reinterpret_cast<__bf16*>(e[1])[o] = (__bf16)x;
        else reinterpret_cast<float*>(e[1])[o] = (flags & 1) ? rnd16(x) : x;
    }
}

int ss_blocks(long nvec) { long b = (nvec + 1023) / 1024; return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
}  // namespace

extern "C" long dkd_sumsq_f32_workspace_bytes(long n) { return (long)ss_blocks(n / 4) * sizeof(float); }

extern "C" int dkd_sumsq_f32(const float* x, long n, float* out2, float* ws, void* stream) {
    if (!x || !out2 || !ws || n <= 0) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(x)) return DKD_ERR_ALIGN;
    const int blocks = ss_blocks(n / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(x), n / 4, ws);
    DKD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, ws, blocks, out2);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_adamw_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, const float* gnorm2, float max_norm, void* p16, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(p) || !dkd_aligned16(g) || !dkd_aligned16(m) || !dkd_aligned16(v)) return DKD_ERR_ALIGN;
    if (((uintptr_t)p16) & 7) return DKD_ERR_ALIGN;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const long nvec = n / 4;
    long b = (nvec + 255) / 256;
    const int blocks = (int)(b > 8192 ? 8192 : b);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(p),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), nvec,
                       lr, beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), gnorm2, max_norm,
                       reinterpret_cast<dkd_bf16x4*>(p16));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_adamw_dev_f32(float* p, const float* g, float* m, float* v, long n, const float* hyper6, const float* gnorm2,
                                 float max_norm, void* p16, void* stream) {
    if (!p || !g || !m || !v || !hyper6 || n <= 0) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(p) || !dkd_aligned16(g) || !dkd_aligned16(m) || !dkd_aligned16(v)) return DKD_ERR_ALIGN;
    if (((uintptr_t)p16) & 7) return DKD_ERR_ALIGN;
    const long nvec = n / 4;
    long b = (nvec + 255) / 256;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((int)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m),
                       reinterpret_cast<float4*>(v), nvec, hyper6, gnorm2, max_norm, reinterpret_cast<dkd_bf16x4*>(p16));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_prep_batch(const void* table, int entries, long max_elems, void* stream) {
    if (!table || entries <= 0 || max_elems <= 0) return DKD_ERR_BAD_ARG;
    long bx = (max_elems + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(prep_batch_kernel, dim3((unsigned)bx, (unsigned)entries), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(table));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_clip_grad_f32(float* g, long n, const float* gnorm2, float max_norm, void* stream) {
    if (!g || !gnorm2 || n <= 0) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(g)) return DKD_ERR_ALIGN;
    const long nvec = n / 4;
    long b = (nvec + 255) / 256;
    hipLaunchKernelGGL(scale_by_clip_kernel, dim3((int)(b > 8192 ? 8192 : b)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(g), nvec, gnorm2, max_norm);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
