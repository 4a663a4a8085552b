// Swin-Transformer student kernels (models/backbones/swin.py of the reference), gfx950.
//
//  * gather_rows      : every token re-ordering of the backbone is ONE row gather with a precomputed int32 map
//                       (zero-pad + cyclic roll + window partition, :283-306; window merge + un-roll + crop, :313-328;
//                       PatchMerging's 2x2 concat with odd-grid padding, :359-371) and so is each of their adjoints
//                       (the maps are injective away from the pad rows).  HBM-bound: reads C floats, writes C floats per row.
//  * winattn fwd/bwd  : window attention with relative-position bias and shift mask (:151-199).  A window is N = 49
//                       tokens x head_dim 32: far too small for an MFMA tiling to pay (49 -> 64 padding, 58 % useful),
//                       and the whole backbone spends ~1 % of the step here, so it is a VALU kernel: one wave per
//                       (window, head), lane = query row (forward, backward phase 1) or key row (backward phase 2), the
//                       other operand broadcast from LDS; the N x N logits never leave registers.
//  * dropout          : counter-based (stateless) Bernoulli mask, regenerated in backward from (seed, offset).
//  * index_add_rows   : deterministic fold of the [N*N, H] bias gradient into the (2w-1)^2-row table.
#include <stdlib.h>

#include "common.h"
#include "dkd_hip.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- row gather ---------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ in, const int* __restrict__ idx, const float* __restrict__ res,
                                   float* __restrict__ out, long rows, int cq) {
    const long total = rows * cq;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / cq;
        const int c = (int)(t - r * cq);
        const int s = idx[r];
        float4 v = s >= 0 ? reinterpret_cast<const float4*>(in)[(long)s * cq + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) {
            const float4 a = reinterpret_cast<const float4*>(res)[t];
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        reinterpret_cast<float4*>(out)[t] = v;
    }
}

// narrow rows (C not a multiple of 4: the [T, heads] relative-position table)
__global__ void gather_rows_scalar_kernel(const float* __restrict__ in, const int* __restrict__ idx, const float* __restrict__ res,
                                          float* __restrict__ out, long rows, int C) {
    const long total = rows * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / C;
        const int s = idx[r];
        float v = s >= 0 ? in[(long)s * C + (t - r * C)] : 0.f;
        if (res) v += res[t];
        out[t] = v;
    }
}

// out[t, :] = sum over rows r with idx[r] == t of src[r, :]   (T and C tiny: (2w-1)^2 x heads)
__global__ void index_add_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out,
                                      int rows, int C) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r)
            if (idx[r] == t) acc += src[(long)r * C + c];
        out[(long)t * C + c] = acc;
    }
}

// dbias[(ij)*H + h] = sum over parts of part[p][h][ij]
__global__ void winattn_dbias_kernel(const float* __restrict__ part, float* __restrict__ out, int parts, int H, int NN) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= H * NN) return;
    const int h = e / NN, ij = e - h * NN;
    float acc = 0.f;
    for (int p = 0; p < parts; ++p) acc += part[((long)p * H + h) * NN + ij];
    out[(long)ij * H + h] = acc;
}

// ---- dropout ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, uint32_t thresh, float inv_keep,
                               uint64_t seed, uint64_t offset) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(seed ^ ((offset + (uint64_t)t) * 0x9E3779B97F4A7C15ull));
        float4 v = reinterpret_cast<const float4*>(x)[t];
        v.x = ((uint32_t)(h & 0xFFFF) >= thresh) ? v.x * inv_keep : 0.f;          // 16 random bits per element
        v.y = ((uint32_t)((h >> 16) & 0xFFFF) >= thresh) ? v.y * inv_keep : 0.f;
        v.z = ((uint32_t)((h >> 32) & 0xFFFF) >= thresh) ? v.z * inv_keep : 0.f;
        v.w = ((uint32_t)(h >> 48) >= thresh) ? v.w * inv_keep : 0.f;
        reinterpret_cast<float4*>(y)[t] = v;
    }
}

// nn.Dropout on a tensor torch autocast keeps in 16 bit (proj_drop / Mlp.drop on bf16 Linear outputs, swin.py:78-85,146,197),
// optionally fused with the residual add that follows it (:331-332): d = rnd(x * mask / (1 - p)) is the 16-bit dropout result,
// y = res + d in y's type (fp32 residual stream: exact fp32 add; bf16 stream: one more rounding).  Same mask function as
// dropout_kernel: element t uses the same 16 random bits whatever the tensor types.
template <typename TX, typename TR, typename TY>
__global__ void dropout_add_io_kernel(const TX* __restrict__ x, const TR* __restrict__ res, TY* __restrict__ y, long n4, uint32_t thresh,
                                      float inv_keep, uint64_t seed, uint64_t offset) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(seed ^ ((offset + (uint64_t)t) * 0x9E3779B97F4A7C15ull));
        float4 v = ::ld4(x, t);
        v.x = ((uint32_t)(h & 0xFFFF) >= thresh) ? v.x * inv_keep : 0.f;
        v.y = ((uint32_t)((h >> 16) & 0xFFFF) >= thresh) ? v.y * inv_keep : 0.f;
        v.z = ((uint32_t)((h >> 32) & 0xFFFF) >= thresh) ? v.z * inv_keep : 0.f;
        v.w = ((uint32_t)(h >> 48) >= thresh) ? v.w * inv_keep : 0.f;
        if (dkd_is16<TX>::value) v = rnd16(v);
        if (res) {
            const float4 a = ::ld4(res, t);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        ::st4(y, t, v);
    }
}

// ---- window attention ---------------------------------------------------------------------------------------------
// One wave per (window, head); a workgroup of 4 waves shares only the head's [N][N] bias tile (LDS, odd pitch: lane-per-row
// and lane-per-column reads are conflict-free).  Lane = query row (forward, backward phase 1) or key row (backward phase 2);
// the OTHER operand's rows are wave-uniform and are read straight from global memory with wave-uniform addresses (one 16-byte
// request per wave-instruction, L1-resident: a window-head's K and V are 12.5 KB).  First version staged them in LDS and read
// them back with same-address ds_read_b128: on gfx950 that "broadcast" costs ~45 LDS cycles per instruction
// (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, scripts/pmc_winattn.sh) and the kernel ran at 5 % of the VALU peak; this form is 2x
// faster and still delivery-bound (a lane-per-row VALU formulation has to broadcast one operand somehow).  The structural fix
// is the MFMA formulation (S^T = K Q^T tiles whose accumulators are the B operand of O^T = V^T P^T, as in attention.hip) with
// the 49 -> 64 padding: ~8k MFMA cycles per window-head against ~130k cycles here; noted in DESIGN.md as next.
constexpr int WA_WAVES = 4;

template <int N>
struct WinCfg {
    static constexpr int NP = N | 1;
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave-uniform rows are read with ordinary vector loads (all 64 lanes, one address).  Routing them through the constant address
// space instead (hipcc then emits s_load_dwordx8 into SGPRs) measured 1.5x SLOWER: ~400 scalar loads per window-head, issued in
// SGPR-sized batches behind s_waitcnt lgkmcnt(0), are latency-bound at 1-2 waves per SIMD.
__device__ __forceinline__ float4 ldu4(const float* p) { return ld4(p); }
__device__ __forceinline__ float lane_value(float v, int lane) {  // lane: wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int N, int HD>
__global__ __launch_bounds__(WA_WAVES * 64) void winattn_fwd_kernel(dkd_winattn_args a) {
    constexpr int NP = WinCfg<N>::NP;
    __shared__ float sBias[N * NP];
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) sBias[(e / N) * NP + (e % N)] = a.bias[(long)e * a.H + h];
    __syncthreads();
    const bool active = lane < N;
    const int li = active ? lane : N - 1;  // idle lanes shadow the last row (loads stay in bounds, nothing is stored)
    for (int w = blockIdx.x * WA_WAVES + wave; w < a.Bw; w += gridDim.x * WA_WAVES) {
        const long row0 = (long)w * N;
        const float* kp = a.k + row0 * a.k_rs + h * HD;  // wave-uniform
        const float* vp = a.v + row0 * a.v_rs + h * HD;
        const int* labp = a.labels ? a.labels + (long)(w % a.nW) * N : nullptr;
        const int my_lab = labp ? labp[li] : 0;
        float q[HD];
        {
            const float* qp = a.q + (row0 + li) * a.q_rs + h * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 t = ld4(qp + d);
                q[d] = t.x * a.scale; q[d + 1] = t.y * a.scale; q[d + 2] = t.z * a.scale; q[d + 3] = t.w * a.scale;
            }
        }
        float s[N];
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float acc0 = sBias[li * NP + j], acc1 = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 8) {
                const float4 k0 = ldu4(kp + (long)j * a.k_rs + d), k1 = ldu4(kp + (long)j * a.k_rs + d + 4);
                acc0 = fmaf(q[d], k0.x, acc0); acc1 = fmaf(q[d + 4], k1.x, acc1);
                acc0 = fmaf(q[d + 1], k0.y, acc0); acc1 = fmaf(q[d + 5], k1.y, acc1);
                acc0 = fmaf(q[d + 2], k0.z, acc0); acc1 = fmaf(q[d + 6], k1.z, acc1);
                acc0 = fmaf(q[d + 3], k0.w, acc0); acc1 = fmaf(q[d + 7], k1.w, acc1);
            }
            float acc = acc0 + acc1;
            if (labp && __builtin_amdgcn_readlane(my_lab, j) != my_lab) acc -= 100.0f;
            s[j] = acc;
            m = fmaxf(m, acc);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            s[j] = __expf(s[j] - m);
            l += s[j];
        }
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 vv = ldu4(vp + (long)j * a.v_rs + d);
                o[d] = fmaf(s[j], vv.x, o[d]); o[d + 1] = fmaf(s[j], vv.y, o[d + 1]);
                o[d + 2] = fmaf(s[j], vv.z, o[d + 2]); o[d + 3] = fmaf(s[j], vv.w, o[d + 3]);
            }
        }
        if (active) {
            const float inv = 1.0f / l;
            float* op = a.o + (row0 + lane) * a.o_rs + h * HD;
#pragma unroll
            for (int d = 0; d < HD; d += 4) st4(op + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv));
        }
    }
}

// Backward.  Phase 1 (lane = query i): recompute the softmax row, delta_i = dO_i . O_i, dQ_i.  Phase 2 (lane = key j): walk the
// queries (their Q / dO rows wave-uniform, their row statistics fetched from lane i with v_readlane), recompute column j of P and
// accumulate dK_j, dV_j in registers and dS into the workgroup's [N][N] bias-gradient tile (one LDS tile per wave, one address per lane: plain read-modify-write).
template <int N, int HD>
__global__ __launch_bounds__(WA_WAVES * 64) void winattn_bwd_kernel(dkd_winattn_args a) {
    constexpr int NP = WinCfg<N>::NP;
    __shared__ float sBias[N * NP];
    __shared__ float sDbAll[WA_WAVES * N * NP];  // one d(bias) tile per wave, folded in a fixed order (bit-reproducible)
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) sBias[(e / N) * NP + (e % N)] = a.bias[(long)e * a.H + h];
    for (int e = threadIdx.x; e < WA_WAVES * N * NP; e += blockDim.x) sDbAll[e] = 0.f;
    float* sDb = sDbAll + wave * N * NP;
    __syncthreads();
    const bool active = lane < N;
    const int li = active ? lane : N - 1;
    for (int w = blockIdx.x * WA_WAVES + wave; w < a.Bw; w += gridDim.x * WA_WAVES) {
        const long row0 = (long)w * N;
        const float* qb = a.q + row0 * a.q_rs + h * HD;    // wave-uniform bases
        const float* kb = a.k + row0 * a.k_rs + h * HD;
        const float* vb = a.v + row0 * a.v_rs + h * HD;
        const float* gb = a.d_o + row0 * a.do_rs + h * HD;
        const int* labp = a.labels ? a.labels + (long)(w % a.nW) * N : nullptr;
        const int my_lab = labp ? labp[li] : 0;
        float m, inv, delta;
        {   // ---- phase 1 ----
            float q[HD], g[HD];
            const float* qp = qb + (long)li * a.q_rs;
            const float* gp = gb + (long)li * a.do_rs;
            const float* op = a.o + (row0 + li) * a.o_rs + h * HD;
            delta = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 t = ld4(qp + d), u = ld4(gp + d), v = ld4(op + d);
                q[d] = t.x * a.scale; q[d + 1] = t.y * a.scale; q[d + 2] = t.z * a.scale; q[d + 3] = t.w * a.scale;
                g[d] = u.x; g[d + 1] = u.y; g[d + 2] = u.z; g[d + 3] = u.w;
                delta += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
            }
            float p[N];
            m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float acc0 = sBias[li * NP + j], acc1 = 0.f;
#pragma unroll
                for (int d = 0; d < HD; d += 8) {
                    const float4 k0 = ldu4(kb + (long)j * a.k_rs + d), k1 = ldu4(kb + (long)j * a.k_rs + d + 4);
                    acc0 = fmaf(q[d], k0.x, acc0); acc1 = fmaf(q[d + 4], k1.x, acc1);
                    acc0 = fmaf(q[d + 1], k0.y, acc0); acc1 = fmaf(q[d + 5], k1.y, acc1);
                    acc0 = fmaf(q[d + 2], k0.z, acc0); acc1 = fmaf(q[d + 6], k1.z, acc1);
                    acc0 = fmaf(q[d + 3], k0.w, acc0); acc1 = fmaf(q[d + 7], k1.w, acc1);
                }
                float acc = acc0 + acc1;
                if (labp && __builtin_amdgcn_readlane(my_lab, j) != my_lab) acc -= 100.0f;
                p[j] = acc;
                m = fmaxf(m, acc);
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                p[j] = __expf(p[j] - m);
                l += p[j];
            }
            inv = 1.0f / l;
            float dq[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float dp0 = 0.f, dp1 = 0.f;
#pragma unroll
                for (int d = 0; d < HD; d += 8) {
                    const float4 v0 = ldu4(vb + (long)j * a.v_rs + d), v1 = ldu4(vb + (long)j * a.v_rs + d + 4);
                    dp0 = fmaf(g[d], v0.x, dp0); dp1 = fmaf(g[d + 4], v1.x, dp1);
                    dp0 = fmaf(g[d + 1], v0.y, dp0); dp1 = fmaf(g[d + 5], v1.y, dp1);
                    dp0 = fmaf(g[d + 2], v0.z, dp0); dp1 = fmaf(g[d + 6], v1.z, dp1);
                    dp0 = fmaf(g[d + 3], v0.w, dp0); dp1 = fmaf(g[d + 7], v1.w, dp1);
                }
                const float ds = p[j] * inv * ((dp0 + dp1) - delta);
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 kk = ldu4(kb + (long)j * a.k_rs + d);
                    dq[d] = fmaf(ds, kk.x, dq[d]); dq[d + 1] = fmaf(ds, kk.y, dq[d + 1]);
                    dq[d + 2] = fmaf(ds, kk.z, dq[d + 2]); dq[d + 3] = fmaf(ds, kk.w, dq[d + 3]);
                }
            }
            if (active) {
                float* dqp = a.dq + (row0 + lane) * a.dq_rs + h * HD;
#pragma unroll
                for (int d = 0; d < HD; d += 4)
                    st4(dqp + d, make_float4(dq[d] * a.scale, dq[d + 1] * a.scale, dq[d + 2] * a.scale, dq[d + 3] * a.scale));
            }
        }
        {   // ---- phase 2 ----
            float kr[HD], vr[HD], dk[HD], dv[HD];
            const float* kp = kb + (long)li * a.k_rs;
            const float* vp = vb + (long)li * a.v_rs;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 t = ld4(kp + d), u = ld4(vp + d);
                kr[d] = t.x; kr[d + 1] = t.y; kr[d + 2] = t.z; kr[d + 3] = t.w;
                vr[d] = u.x; vr[d + 1] = u.y; vr[d + 2] = u.z; vr[d + 3] = u.w;
                dk[d] = dk[d + 1] = dk[d + 2] = dk[d + 3] = 0.f;
                dv[d] = dv[d + 1] = dv[d + 2] = dv[d + 3] = 0.f;
            }
#pragma unroll 1
            for (int i = 0; i < N; ++i) {
                float s0 = 0.f, s1 = 0.f, dp0 = 0.f, dp1 = 0.f;
                float qq[HD], gg[HD];  // row i of Q and of dO: wave-uniform loads
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 t = ldu4(qb + (long)i * a.q_rs + d), u = ldu4(gb + (long)i * a.do_rs + d);
                    qq[d] = t.x; qq[d + 1] = t.y; qq[d + 2] = t.z; qq[d + 3] = t.w;
                    gg[d] = u.x; gg[d + 1] = u.y; gg[d + 2] = u.z; gg[d + 3] = u.w;
                }
#pragma unroll
                for (int d = 0; d < HD; d += 2) {
                    s0 = fmaf(kr[d], qq[d], s0); s1 = fmaf(kr[d + 1], qq[d + 1], s1);
                    dp0 = fmaf(vr[d], gg[d], dp0); dp1 = fmaf(vr[d + 1], gg[d + 1], dp1);
                }
                float sc = fmaf(s0 + s1, a.scale, sBias[i * NP + li]);
                if (labp && __builtin_amdgcn_readlane(my_lab, i) != my_lab) sc -= 100.0f;
                const float p = __expf(sc - lane_value(m, i)) * lane_value(inv, i);
                const float ds = p * ((dp0 + dp1) - lane_value(delta, i));
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dk[d] = fmaf(ds, qq[d], dk[d]);
                    dv[d] = fmaf(p, gg[d], dv[d]);
                }
                if (active) sDb[i * NP + lane] += ds;
            }
            if (active) {
                float* dkp = a.dk + (row0 + lane) * a.dk_rs + h * HD;
                float* dvp = a.dv + (row0 + lane) * a.dv_rs + h * HD;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    st4(dkp + d, make_float4(dk[d] * a.scale, dk[d + 1] * a.scale, dk[d + 2] * a.scale, dk[d + 3] * a.scale));
                    st4(dvp + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]));
                }
            }
        }
    }
    __syncthreads();
    float* part = a.dbias_part + ((long)blockIdx.x * a.H + h) * N * N;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int o = (e / N) * NP + (e % N);
        float t = sDbAll[o];
#pragma unroll
        for (int wv = 1; wv < WA_WAVES; ++wv) t += sDbAll[wv * N * NP + o];
        part[e] = t;
    }
}

int check_args(const dkd_winattn_args* a, bool bwd) {
    if (!a || !a->q || !a->k || !a->v || !a->o || !a->bias || a->Bw <= 0 || a->H <= 0 || a->nW <= 0) return DKD_ERR_BAD_ARG;
    if (a->labels && a->Bw % a->nW) return DKD_ERR_BAD_ARG;
    if (bwd && (!a->d_o || !a->dq || !a->dk || !a->dv || !a->dbias_part)) return DKD_ERR_BAD_ARG;
    const int64_t rs[] = {a->q_rs, a->k_rs, a->v_rs, a->o_rs};
    for (int64_t s : rs)
        if (s < (int64_t)a->H * a->hd || (s & 3)) return DKD_ERR_ALIGN;
    if (bwd) {
        const int64_t rb[] = {a->do_rs, a->dq_rs, a->dk_rs, a->dv_rs};
        for (int64_t s : rb)
            if (s < (int64_t)a->H * a->hd || (s & 3)) return DKD_ERR_ALIGN;
        if (!dkd_aligned16(a->d_o) || !dkd_aligned16(a->dq) || !dkd_aligned16(a->dk) || !dkd_aligned16(a->dv)) return DKD_ERR_ALIGN;
    }
    if (!dkd_aligned16(a->q) || !dkd_aligned16(a->k) || !dkd_aligned16(a->v) || !dkd_aligned16(a->o)) return DKD_ERR_ALIGN;
    return DKD_OK;
}

template <int N, int HD>
int launch_fwd(const dkd_winattn_args* a, hipStream_t st) {
    dim3 grid(dkd_winattn_parts(a->Bw, a->H), a->H);
    hipLaunchKernelGGL((winattn_fwd_kernel<N, HD>), grid, dim3(WA_WAVES * 64), 0, st, *a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
template <int N, int HD>
int launch_bwd(const dkd_winattn_args* a, hipStream_t st) {
    dim3 grid(dkd_winattn_parts(a->Bw, a->H), a->H);
    hipLaunchKernelGGL((winattn_bwd_kernel<N, HD>), grid, dim3(WA_WAVES * 64), 0, st, *a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

}  // namespace

// matrix-core kernels (swin_mfma.hip): the default; DKD_WINATTN_VALU=1 selects the lane-per-row kernels above for A/B runs
int dkd_winattn_mfma_fwd(const dkd_winattn_args* a, hipStream_t st);
int dkd_winattn_mfma_bwd(const dkd_winattn_args* a, hipStream_t st);
static bool winattn_use_valu() {
    static const bool v = getenv("DKD_WINATTN_VALU") != nullptr && getenv("DKD_WINATTN_VALU")[0] == '1';
    return v;
}

extern "C" int dkd_winattn_parts(int Bw, int H) {
    if (Bw <= 0 || H <= 0) return 0;
    const int want = (768 + H - 1) / H;  // ~3 workgroups per CU over (parts x heads)
    const int need = (Bw + WA_WAVES - 1) / WA_WAVES;
    return need < want ? need : want;
}

extern "C" int dkd_winattn_fwd_f32(const dkd_winattn_args* a, void* stream) {
    const int rc = check_args(a, false);
    if (rc) return rc;
    if (a->N == 49 && (a->hd == 32 || a->hd == 16) && (a->policy16 || !winattn_use_valu())) return dkd_winattn_mfma_fwd(a, (hipStream_t)stream);
    if (a->policy16) return DKD_ERR_UNSUPPORTED;
    if (a->N == 49 && a->hd == 32) return launch_fwd<49, 32>(a, (hipStream_t)stream);
    if (a->N == 49 && a->hd == 16) return launch_fwd<49, 16>(a, (hipStream_t)stream);
    return DKD_ERR_UNSUPPORTED;
}

extern "C" int dkd_winattn_bwd_f32(const dkd_winattn_args* a, void* stream) {
    const int rc = check_args(a, true);
    if (rc) return rc;
    if (a->parts != dkd_winattn_parts(a->Bw, a->H)) return DKD_ERR_BAD_ARG;
    if (a->N == 49 && (a->hd == 32 || a->hd == 16) && (a->policy16 || !winattn_use_valu())) return dkd_winattn_mfma_bwd(a, (hipStream_t)stream);
    if (a->policy16) return DKD_ERR_UNSUPPORTED;
    if (a->N == 49 && a->hd == 32) return launch_bwd<49, 32>(a, (hipStream_t)stream);
    if (a->N == 49 && a->hd == 16) return launch_bwd<49, 16>(a, (hipStream_t)stream);
    return DKD_ERR_UNSUPPORTED;
}

extern "C" int dkd_winattn_dbias_f32(const float* part, float* dbias, int parts, int H, int N, void* stream) {
    if (!part || !dbias || parts <= 0 || H <= 0 || N <= 0) return DKD_ERR_BAD_ARG;
    const int total = H * N * N;
    hipLaunchKernelGGL(winattn_dbias_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, dbias, parts, H,
                       N * N);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_gather_rows_f32(const float* in, const int* idx, const float* res, float* out, long rows, int C, void* stream) {
    if (!in || !idx || !out || rows <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || !dkd_aligned16(in) || !dkd_aligned16(out) || (res && !dkd_aligned16(res))) {
        const long total = rows * C, blocks = (total + 255) / 256;
        hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                           (hipStream_t)stream, in, idx, res, out, rows, C);
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
    const long total = rows * (C / 4);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, in,
                       idx, res, out, rows, C / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_index_add_rows_f32(const float* src, const int* idx, float* out, int rows, int T, int C, void* stream) {
    if (!src || !idx || !out || rows <= 0 || T <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(index_add_rows_kernel, dim3(T), dim3(64), 0, (hipStream_t)stream, src, idx, out, rows, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dropout_add_io(const void* x, int x16, const void* res, int res16, void* y, int y16, long n, float p, int64_t seed,
                                  int64_t offset, void* stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f)) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(x) || !dkd_aligned16(y) || (res && !dkd_aligned16(res))) return DKD_ERR_ALIGN;
    const uint32_t thresh = (uint32_t)(p * 65536.0f + 0.5f);
    const long n4 = n / 4;
    const long blocks = (n4 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    DKD_IO_TYPE(x16, TX, DKD_IO_TYPE(res16, TR, DKD_IO_TYPE(y16, TY,
        hipLaunchKernelGGL((dropout_add_io_kernel<TX, TR, TY>), grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const TX*>(x),
                           reinterpret_cast<const TR*>(res), reinterpret_cast<TY*>(y), n4, thresh, 1.0f / (1.0f - p), (uint64_t)seed,
                           (uint64_t)offset))));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_dropout_f32(const float* x, float* y, long n, float p, int64_t seed, int64_t offset, void* stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f)) return DKD_ERR_BAD_ARG;
    if ((n & 3) || !dkd_aligned16(x) || !dkd_aligned16(y)) return DKD_ERR_ALIGN;
    const uint32_t thresh = (uint32_t)(p * 65536.0f + 0.5f);
    const long n4 = n / 4;
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, x, y,
                       n4, thresh, 1.0f / (1.0f - p), (uint64_t)seed, (uint64_t)offset);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
