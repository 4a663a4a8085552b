// ScaleKD loss tail (losses/scalekd.py:67-127): per-token L2 normalisation of student and teacher features,
// sum-of-squares distance and cosine similarity, forward and analytic backward, one wavefront per token row.
// Both maps are token-major [B*HW, D] (the teacher's [B,D,P,P] feature map is a permuted view of exactly this
// memory, models/backbones/dinov2.py:40), so "normalize over dim=1 of NCHW" is a contiguous row reduction.
// The frequency variant (DCT -> zero DC -> IDCT, :107-118) equals removing each (image, channel) map's spatial
// mean; the means arrive as [B, D] vectors (dkd_colreduce_f32) and are subtracted on the fly.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "dkd_hip.h"

namespace {
constexpr int KD_MAXV = 8;  // float4 per lane: D <= 2048
constexpr int KD_U = 1;     // row groups per wave and pass (2 measured no better: hipcc sinks the second group's loads below the first reduction)

// Rows narrower than 64 float4 share a wave (LPR lanes per row, 64 / LPR rows per wave), VPL float4 per lane; with KD_U = 2 a
// lane has 2 * 2 * VPL 16-byte loads in flight (D = 384: 12) instead of the 4 of the one-row-per-wave version, which ran at
// 2.7 TB/s because of it (traffic == algorithmic bytes, so latency-bound: profiles/r01_pmc_traffic.md).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int VPL>
struct RowVals {
    float4 s[VPL], t[VPL];
};

// Straight-line loads: out-of-range lanes / rows read a clamped (valid) address and are zeroed by a select afterwards, and the
// frequency variant is a template flag - with run-time "if (ms)" branches hipcc put an s_waitcnt vmcnt(0) behind every mean
// load, which serialised the row loads (seen in the ISA); now every load of a row group is issued before the first use.
template <int LPR, int VPL, bool FREQ>
__device__ __forceinline__ void load_row(RowVals<VPL>& v, const float* __restrict__ S, const float* __restrict__ T,
                                         const float* __restrict__ ms, const float* __restrict__ mt, long row, long rows,
                                         long rows_per_img, int D, int lir) {
    const int nv = D >> 2;
    const bool ok = row < rows;
    const long r = ok ? row : rows - 1;
    const float4* sr = reinterpret_cast<const float4*>(S + r * D);
    const float4* tr = reinterpret_cast<const float4*>(T + r * D);
    float4 m_s[VPL], m_t[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = min(lir + i * LPR, nv - 1);
        v.s[i] = sr[c];
        v.t[i] = tr[c];
    }
    if (FREQ) {
        const long b = r / rows_per_img;
        const float4* msr = reinterpret_cast<const float4*>(ms + b * D);
        const float4* mtr = reinterpret_cast<const float4*>(mt + b * D);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = min(lir + i * LPR, nv - 1);
            m_s[i] = msr[c];
            m_t[i] = mtr[c];
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bool live = ok && lir + i * LPR < nv;
        float4 a = v.s[i], d = v.t[i];
        if (FREQ) {
            a.x -= m_s[i].x; a.y -= m_s[i].y; a.z -= m_s[i].z; a.w -= m_s[i].w;
            d.x -= m_t[i].x; d.y -= m_t[i].y; d.z -= m_t[i].z; d.w -= m_t[i].w;
        }
        v.s[i] = live ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        v.t[i] = live ? d : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// Forward: per-workgroup partial sums -> ws, folded in fp64 by kd_loss_final_kernel (second launch).  NW = 16 waves per
// workgroup at full size: the same 8 waves per SIMD in a quarter of the workgroups, i.e. a quarter of the partials to fold.
// A single-launch form (last-arriving workgroup folds: sc1 partial stores, s_waitcnt, one device-scope ticket atomic per
// workgroup, sc1 loads) was built and measured in round 2 and is NOT kept: the ~12 ns per atomic serialise at the memory side
// when all workgroups finish together - 2048 arrivals 36 us, 512 arrivals 19.2 us, against 15.9 us for the two launches
// (bs 64, D 384, scripts/bench_loss.py, gpurun_out/r2d_loss*.log).
template <int LPR, int VPL, bool FREQ, int NW>
__global__ __launch_bounds__(NW * 64) void kd_loss_fwd_kernel(
    const float* __restrict__ S, const float* __restrict__ T, const float* __restrict__ ms, const float* __restrict__ mt,
    float* __restrict__ partials, long rows, long rows_per_img, int D) {
    constexpr int RPW = 64 / LPR;
    __shared__ float red[2][NW];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane / LPR, lir = lane % LPR;
    float acc_sq = 0.f, acc_cos = 0.f;
    const long stride = (long)gridDim.x * NW * RPW;
    for (long row0 = ((long)blockIdx.x * NW + wid) * RPW + sub; row0 < rows; row0 += stride * KD_U) {
        RowVals<VPL> v[KD_U];
#pragma unroll
        for (int u = 0; u < KD_U; ++u) load_row<LPR, VPL, FREQ>(v[u], S, T, ms, mt, row0 + u * stride, rows, rows_per_img, D, lir);
#pragma unroll
        for (int u = 0; u < KD_U; ++u) {
            float ss = 0.f, tt = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                ss += dot4(v[u].s[i], v[u].s[i]);
                tt += dot4(v[u].t[i], v[u].t[i]);
            }
            ss = group_sum<LPR>(ss);
            tt = group_sum<LPR>(tt);
            const float ns = sqrtf(ss), nt = sqrtf(tt);
            const float is = 1.f / fmaxf(ns, 1e-12f), it = 1.f / fmaxf(nt, 1e-12f);  // F.normalize eps
            float sq = 0.f, dt = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float4 a = make_float4(v[u].s[i].x * is, v[u].s[i].y * is, v[u].s[i].z * is, v[u].s[i].w * is);
                const float4 b = make_float4(v[u].t[i].x * it, v[u].t[i].y * it, v[u].t[i].z * it, v[u].t[i].w * it);
                const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
                sq += dot4(d, d);
                dt += dot4(a, b);
            }
            sq = group_sum<LPR>(sq);
            dt = group_sum<LPR>(dt);
            // cosine_similarity of the normalised maps: norms of s^ and t^ clamped at 1e-8
            const float nsh = ns * is, nth = nt * it;
            if (lir == 0 && row0 + u * stride < rows) {
                acc_sq += sq;
                acc_cos += dt / (fmaxf(nsh, 1e-8f) * fmaxf(nth, 1e-8f));
            }
        }
    }
    // rows of one wave sit in lanes 0, LPR, 2 LPR ...: fold them in lane order
    if (RPW > 1) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            acc_sq += __shfl_xor(acc_sq, o, 64);
            acc_cos += __shfl_xor(acc_cos, o, 64);
        }
    }
    if (lane == 0) {
        red[0][wid] = acc_sq;
        red[1][wid] = acc_cos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float p0 = 0.f, p1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            p0 += red[0][w];
            p1 += red[1][w];
        }
        partials[2 * blockIdx.x + 0] = p0;
        partials[2 * blockIdx.x + 1] = p1;
    }
}

// second launch: one workgroup folds the per-workgroup partials in fp64, in index order (deterministic)
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
template <int LPR, int VPL, bool FREQ>
__global__ __launch_bounds__(256) void kd_loss_bwd_kernel(const float* __restrict__ S, const float* __restrict__ T,
                                                          const float* __restrict__ ms, const float* __restrict__ mt,
                                                          const float* __restrict__ gout, float* __restrict__ dS,
                                                          long rows, long rows_per_img, int D, float coef) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane / LPR, lir = lane % LPR;
    const float c2 = 2.f * coef * (gout ? gout[0] : 1.f);
    const int nv = D >> 2;
    const long stride = (long)gridDim.x * 4 * RPW;
    for (long row0 = ((long)blockIdx.x * 4 + wid) * RPW + sub; row0 < rows; row0 += stride * KD_U) {
        RowVals<VPL> v[KD_U];
#pragma unroll
        for (int u = 0; u < KD_U; ++u) load_row<LPR, VPL, FREQ>(v[u], S, T, ms, mt, row0 + u * stride, rows, rows_per_img, D, lir);
#pragma unroll
        for (int u = 0; u < KD_U; ++u) {
            const long row = row0 + u * stride;
            float ss = 0.f, tt = 0.f, st = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                ss += dot4(v[u].s[i], v[u].s[i]);
                tt += dot4(v[u].t[i], v[u].t[i]);
                st += dot4(v[u].s[i], v[u].t[i]);
            }
            ss = group_sum<LPR>(ss);
            tt = group_sum<LPR>(tt);
            st = group_sum<LPR>(st);
            const float ns = sqrtf(ss), nt = sqrtf(tt);
            const float is = 1.f / fmaxf(ns, 1e-12f), it = 1.f / fmaxf(nt, 1e-12f);
            // s^.g = c2 (s^.s^ - s^.t^)
            const float shsh = ss * is * is, shth = st * is * it;
            const bool clamped = ns <= 1e-12f;
            const float proj = clamped ? 0.f : c2 * (shsh - shth);
            if (row >= rows) continue;
            float4* dr = reinterpret_cast<float4*>(dS + row * D);
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = lir + i * LPR;
                if (c < nv) {
                    const float4 a = make_float4(v[u].s[i].x * is, v[u].s[i].y * is, v[u].s[i].z * is, v[u].s[i].w * is);
                    const float4 b = make_float4(v[u].t[i].x * it, v[u].t[i].y * it, v[u].t[i].z * it, v[u].t[i].w * it);
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
}

// (LPR, VPL) for a row of nv float4: the narrowest lane group that holds it in <= 3 float4 per lane, else a whole wave
#define DKD_KD_DISPATCH(nv, CALL)          \
    do {                                   \
        if ((nv) <= 48) { CALL(16, 3); }   \
        else if ((nv) <= 96) { CALL(32, 3); } \
        else if ((nv) <= 192) { CALL(64, 3); } \
        else if ((nv) <= 256) { CALL(64, 4); } \
        else if ((nv) <= 384) { CALL(64, 6); } \
        else { CALL(64, 8); }              \
    } while (0)

int kd_rows_per_block(int nv) { return 4 * (nv <= 48 ? 4 : (nv <= 96 ? 2 : 1)); }
constexpr int KD_MAX_BLOCKS = 2048;
// Every row in flight at once where the chip can hold it: one row group per wave, but never more workgroups than are resident
// together (a partial second round of a 10 us kernel costs a whole round) - the kernel's own occupancy, asked once per variant.
template <auto kernel, int NW = 4>  // (a non-type parameter: instantiations of one signature must not share the cached value)
int kd_blocks(long rows, int D) {
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, NW * 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        resident = per_cu * prop.multiProcessorCount;
        if (resident > KD_MAX_BLOCKS) resident = KD_MAX_BLOCKS;
    }
    const long per = (long)kd_rows_per_block(D >> 2) * KD_U * (NW / 4);
    const long b = (rows + per - 1) / per;
    return (int)(b < resident ? b : resident);
}
// the scalar tail of _compute_losses (train/distillation_module.py:238-245) in one launch: up to 8 loss terms, each its own
// one-element tensor (pointers by value: no device-side table), weighted[i] = w[i] t[i], pairs[l] = weighted[2l] + weighted[2l+1],
// total = sum_l pairs[l] in index order; backward g[i] = w[i] (gtotal + gpairs[i / 2] + gweighted[i]) (absent terms NULL)
struct LossTerms { const float* t[8]; };
__global__ void loss_combine_fwd_kernel(LossTerms in, const float* __restrict__ w, int n, float* __restrict__ weighted,
                                        float* __restrict__ pairs, float* __restrict__ total) {
    if (threadIdx.x != 0) return;
    float tot = 0.f;
    for (int l = 0; l < n / 2; ++l) {
        const float a = in.t[2 * l][0] * w[2 * l], b = in.t[2 * l + 1][0] * w[2 * l + 1];
        weighted[2 * l] = a;
        weighted[2 * l + 1] = b;
        const float p = a + b;
        pairs[l] = p;
        tot += p;
    }
    *total = tot;
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ w, const float* __restrict__ gw, const float* __restrict__ gpairs,
                                        const float* __restrict__ gtotal, int n, float* __restrict__ g) {
    const int i = threadIdx.x;
    if (i >= n) return;
    float v = 0.f;   // (the order autograd's chain of adds takes: total, then pairs, then weighted)
    bool any = false;
    if (gtotal) { v = w[i] * gtotal[0]; any = true; }
    if (gpairs) { const float t = w[i] * gpairs[i / 2]; v = any ? v + t : t; any = true; }
    if (gw) { const float t = w[i] * gw[i]; v = any ? v + t : t; }
    g[i] = v;
}
}  // namespace

extern "C" int dkd_loss_combine_fwd_f32(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4,
                                        const float* t5, const float* t6, const float* t7, const float* w, int n, float* weighted,
                                        float* pairs, float* total, void* stream) {
    if (!w || !weighted || !pairs || !total || n <= 0 || n > 8 || (n & 1)) return DKD_ERR_BAD_ARG;
    LossTerms in{{t0, t1, t2, t3, t4, t5, t6, t7}};
    for (int i = 0; i < n; ++i)
        if (!in.t[i]) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, w, n, weighted, pairs, total);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_loss_combine_bwd_f32(const float* w, const float* gweighted, const float* gpairs, const float* gtotal, int n, float* g,
                                        void* stream) {
    if (!w || !g || n <= 0 || n > 8 || (!gweighted && !gpairs && !gtotal)) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, gweighted, gpairs, gtotal, n, g);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" long dkd_kd_loss_f32_workspace_bytes(long rows) { return 2L * KD_MAX_BLOCKS * sizeof(float); }

extern "C" int dkd_kd_loss_fwd_f32(const float* S, const float* T, const float* mean_s, const float* mean_t, float* out2,
                                   float* ws, long rows, long rows_per_img, int D, float alpha, void* stream) {
    if (!S || !T || !out2 || !ws || rows <= 0 || rows_per_img <= 0 || rows % rows_per_img) return DKD_ERR_BAD_ARG;
    if ((D & 3) || D > KD_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int nv = D >> 2;
    hipStream_t st = (hipStream_t)stream;
    const float batch = (float)(rows / rows_per_img);
    if ((mean_s == nullptr) != (mean_t == nullptr)) return DKD_ERR_BAD_ARG;  // the frequency variant centres BOTH maps
    static const bool small_wg = getenv("DKD_KD_WAVES") && !strcmp(getenv("DKD_KD_WAVES"), "4");  // A/B (scripts/bench_loss.py)
    const bool big = !small_wg && rows >= 4096;
    int blocks = 0;
#define DKD_KD_FWD1(L, V, F, W)                                                                                               \
    do {                                                                                                                      \
        blocks = kd_blocks<kd_loss_fwd_kernel<L, V, F, W>, W>(rows, D);                                                       \
        hipLaunchKernelGGL((kd_loss_fwd_kernel<L, V, F, W>), dim3(blocks), dim3(W * 64), 0, st, S, T, mean_s, mean_t, ws,    \
                           rows, rows_per_img, D);                                                                           \
    } while (0)
#define DKD_KD_FWD(L, V)                                   \
    if (mean_s) {                                          \
        if (big && V <= 4) DKD_KD_FWD1(L, V, true, (V <= 4 ? 16 : 4)); /* (wide rows: 1024 threads would spill) */ \
        else DKD_KD_FWD1(L, V, true, 4);                   \
    } else {                                               \
        if (big && V <= 4) DKD_KD_FWD1(L, V, false, (V <= 4 ? 16 : 4)); \
        else DKD_KD_FWD1(L, V, false, 4);                  \
    }
    DKD_KD_DISPATCH(nv, DKD_KD_FWD);
#undef DKD_KD_FWD1
#undef DKD_KD_FWD
    DKD_LAUNCH_CHECK();
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
    const int nv = D >> 2;
    if ((mean_s == nullptr) != (mean_t == nullptr)) return DKD_ERR_BAD_ARG;
#define DKD_KD_BWD(L, V)                                                                                                           \
    if (mean_s)                                                                                                                    \
        hipLaunchKernelGGL((kd_loss_bwd_kernel<L, V, true>), dim3(kd_blocks<kd_loss_bwd_kernel<L, V, true>>(rows, D)), dim3(256), \
                           0, (hipStream_t)stream, S, T, mean_s, mean_t, grad_out, dS, rows, rows_per_img, D, alpha / batch);      \
    else                                                                                                                           \
        hipLaunchKernelGGL((kd_loss_bwd_kernel<L, V, false>), dim3(kd_blocks<kd_loss_bwd_kernel<L, V, false>>(rows, D)),          \
                           dim3(256), 0, (hipStream_t)stream, S, T, mean_s, mean_t, grad_out, dS, rows, rows_per_img, D,           \
                           alpha / batch)
    DKD_KD_DISPATCH(nv, DKD_KD_BWD);
#undef DKD_KD_BWD
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
