// ScaleKD loss tail (losses/scalekd.py:67-127): per-token L2 normalisation of student and teacher features,
// sum-of-squares distance and cosine similarity, forward and analytic backward, one wavefront per token row.
// Both maps are token-major [B*HW, D] (the teacher's [B,D,P,P] feature map is a permuted view of exactly this
// memory, models/backbones/dinov2.py:40), so "normalize over dim=1 of NCHW" is a contiguous row reduction.
// The frequency variant (DCT -> zero DC -> IDCT, :107-118) equals removing each (image, channel) map's spatial
// mean; the means arrive as [B, D] vectors (dkd_colreduce_f32) and are subtracted on the fly.
#include "common.h"
#include "dkd_hip.h"

namespace {
constexpr int KD_MAXV = 8;  // float4 per lane: D <= 2048

struct RowVals {
    float4 s[KD_MAXV], t[KD_MAXV];
};

__device__ __forceinline__ void load_row(RowVals& v, const float* __restrict__ S, const float* __restrict__ T,
                                         const float* __restrict__ ms, const float* __restrict__ mt, long row, long b,
                                         int D, int lane) {
    const int nv = D >> 2;
    const float4* sr = reinterpret_cast<const float4*>(S + row * D);
    const float4* tr = reinterpret_cast<const float4*>(T + row * D);
#pragma unroll
    for (int i = 0; i < KD_MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            float4 a = sr[c], d = tr[c];
            if (ms) {
                const float4 m = reinterpret_cast<const float4*>(ms + b * D)[c];
                a.x -= m.x; a.y -= m.y; a.z -= m.z; a.w -= m.w;
            }
            if (mt) {
                const float4 m = reinterpret_cast<const float4*>(mt + b * D)[c];
                d.x -= m.x; d.y -= m.y; d.z -= m.z; d.w -= m.w;
            }
            v.s[i] = a;
            v.t[i] = d;
        } else {
            v.s[i] = v.t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

__global__ __launch_bounds__(256) void kd_loss_fwd_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                          const float* __restrict__ ms, const float* __restrict__ mt,
                                                          float* __restrict__ partials, long rows, long rows_per_img,
                                                          int D) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc_sq = 0.f, acc_cos = 0.f;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        RowVals v;
        load_row(v, S, T, ms, mt, row, row / rows_per_img, D, lane);
        float ss = 0.f, tt = 0.f;
#pragma unroll
        for (int i = 0; i < KD_MAXV; ++i) {
            ss += dot4(v.s[i], v.s[i]);
            tt += dot4(v.t[i], v.t[i]);
        }
        ss = wave_sum(ss);
        tt = wave_sum(tt);
        const float ns = sqrtf(ss), nt = sqrtf(tt);
        const float is = 1.f / fmaxf(ns, 1e-12f), it = 1.f / fmaxf(nt, 1e-12f);  // F.normalize eps
        float sq = 0.f, dt = 0.f;
#pragma unroll
        for (int i = 0; i < KD_MAXV; ++i) {
            const float4 a = make_float4(v.s[i].x * is, v.s[i].y * is, v.s[i].z * is, v.s[i].w * is);
            const float4 b = make_float4(v.t[i].x * it, v.t[i].y * it, v.t[i].z * it, v.t[i].w * it);
            const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
            sq += dot4(d, d);
            dt += dot4(a, b);
        }
        sq = wave_sum(sq);
        dt = wave_sum(dt);
        // cosine_similarity of the normalised maps: norms of s^ and t^ clamped at 1e-8
        const float nsh = ns * is, nth = nt * it;
        acc_sq += sq;
        acc_cos += dt / (fmaxf(nsh, 1e-8f) * fmaxf(nth, 1e-8f));
    }
    if (lane == 0) {
        red[0][wid] = acc_sq;
        red[1][wid] = acc_cos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void kd_loss_final_kernel(const float* __restrict__ partials, int n, float* __restrict__ out,
                                                            float loss_scale, float sim_scale) {
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += (double)partials[2 * i];
        b += (double)partials[2 * i + 1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (double)loss_scale);
        out[1] = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (double)sim_scale);
    }
}

// d loss / d s (centred s when ms != null):  g = 2*coef*gout*(s^ - t^);  ds = (g - s^ (s^ . g)) / |s|
__global__ __launch_bounds__(256) void kd_loss_bwd_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                          const float* __restrict__ ms, const float* __restrict__ mt,
                                                          const float* __restrict__ gout, float* __restrict__ dS,
                                                          long rows, long rows_per_img, int D, float coef) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float c2 = 2.f * coef * (gout ? gout[0] : 1.f);
    const int nv = D >> 2;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        RowVals v;
        load_row(v, S, T, ms, mt, row, row / rows_per_img, D, lane);
        float ss = 0.f, tt = 0.f, st = 0.f;
#pragma unroll
        for (int i = 0; i < KD_MAXV; ++i) {
            ss += dot4(v.s[i], v.s[i]);
            tt += dot4(v.t[i], v.t[i]);
            st += dot4(v.s[i], v.t[i]);
        }
        ss = wave_sum(ss);
        tt = wave_sum(tt);
        st = wave_sum(st);
        const float ns = sqrtf(ss), nt = sqrtf(tt);
        const float is = 1.f / fmaxf(ns, 1e-12f), it = 1.f / fmaxf(nt, 1e-12f);
        // s^.g = c2 (s^.s^ - s^.t^)
        const float shsh = ss * is * is, shth = st * is * it;
        const bool clamped = ns <= 1e-12f;
        const float proj = clamped ? 0.f : c2 * (shsh - shth);
        float4* dr = reinterpret_cast<float4*>(dS + row * D);
#pragma unroll
        for (int i = 0; i < KD_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float4 a = make_float4(v.s[i].x * is, v.s[i].y * is, v.s[i].z * is, v.s[i].w * is);
                const float4 b = make_float4(v.t[i].x * it, v.t[i].y * it, v.t[i].z * it, v.t[i].w * it);
                float4 o;
                o.x = (c2 * (a.x - b.x) - a.x * proj) * is;
                o.y = (c2 * (a.y - b.y) - a.y * proj) * is;
                o.z = (c2 * (a.z - b.z) - a.z * proj) * is;
                o.w = (c2 * (a.w - b.w) - a.w * proj) * is;
                dr[c] = o;
            }
        }
    }
}

int kd_blocks(long rows) { return (int)((rows + 3) / 4 < 1024 ? (rows + 3) / 4 : 1024); }
}  // namespace

extern "C" long dkd_kd_loss_f32_workspace_bytes(long rows) { return (long)kd_blocks(rows) * 2 * sizeof(float); }

extern "C" int dkd_kd_loss_fwd_f32(const float* S, const float* T, const float* mean_s, const float* mean_t, float* out2,
                                   float* ws, long rows, long rows_per_img, int D, float alpha, void* stream) {
    if (!S || !T || !out2 || !ws || rows <= 0 || rows_per_img <= 0 || rows % rows_per_img) return DKD_ERR_BAD_ARG;
    if ((D & 3) || D > KD_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int blocks = kd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(kd_loss_fwd_kernel, dim3(blocks), dim3(256), 0, st, S, T, mean_s, mean_t, ws, rows, rows_per_img, D);
    DKD_LAUNCH_CHECK();
    const float batch = (float)(rows / rows_per_img);
    hipLaunchKernelGGL(kd_loss_final_kernel, dim3(1), dim3(256), 0, st, ws, blocks, out2, alpha / batch, 1.0f / (float)rows);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_kd_loss_bwd_f32(const float* S, const float* T, const float* mean_s, const float* mean_t,
                                   const float* grad_out, float* dS, long rows, long rows_per_img, int D, float alpha,
                                   void* stream) {
    if (!S || !T || !dS || rows <= 0 || rows_per_img <= 0 || rows % rows_per_img) return DKD_ERR_BAD_ARG;
    if ((D & 3) || D > KD_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const float batch = (float)(rows / rows_per_img);
    hipLaunchKernelGGL(kd_loss_bwd_kernel, dim3(kd_blocks(rows)), dim3(256), 0, (hipStream_t)stream, S, T, mean_s, mean_t,
                       grad_out, dS, rows, rows_per_img, D, alpha / batch);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
