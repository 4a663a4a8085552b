// Swin window attention on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32), gfx950.
//
// One wave per (window, head); the 49 tokens are padded to 64 = 4 tiles of 16.  Nothing is broadcast: every MFMA operand is a
// per-lane fragment read straight from global memory (a float4 of 4 consecutive head-dim elements of "its" row, or 4 scalars
// of 4 consecutive rows), and - the trick of attention.hip - the accumulators of one product ARE the B operand of the next:
//
//   S^T[j][i] = sum_d K[j][d] Q[i][d]        D rows = key j = 4g+r, cols = query i = lane%16          ("T layout")
//   O^T[d][i] = sum_j V[j][d] P^T[j][i]      B operand wants (k = j = 4g+s, col = i = lane%16): exactly the S^T accumulators
//
// with lane = 16 g + c.  The k-index of an MFMA step is permuted (lane group g supplies k = 4g+s at step s) identically for
// both operands, as in the GEMM.  Softmax runs on the accumulator registers: a query's 64 logits sit in 16 registers of each
// of the 4 lanes {c, c+16, c+32, c+48} -> in-lane reductions plus two xor-shuffles.
// The backward pass needs the logits in both orientations (dQ contracts over keys, dK / dV over queries), so it recomputes
// them twice, once per layout, from the same fragments with the operand roles swapped; row statistics are recomputed per
// layout (layout "N": 16-lane DPP reductions).  d(bias) is accumulated into the workgroup's LDS tile from the T layout.
//
// MFMA work per (window, head): forward 256 issues (8 k-cycles), backward 896: ~8 k / ~29 k cycles against ~130 k / ~330 k
// for the lane-per-row VALU kernels of swin.hip, which stay selectable (DKD_WINATTN_VALU=1) as the A/B reference.
//
// T = __bf16 (dkd_winattn_args.policy16): the same kernels as torch autocast runs WindowAttention.forward (swin.py:151-199) - q, k,
// v, o and their gradients are bf16 in HBM; q * scale, q k^T, P v and every matmul of the backward pass round their result to bf16
// (fp32 MFMA on bf16 values: exact products, fp32 accumulation, one rounding - what a bf16 matmul delivers); adding the fp32
// bias table promotes the logits to fp32, the softmax and its backward are fp32, the probabilities are rounded for P v, and
// d(bias) is the fp32 softmax gradient summed over windows.
#include "common.h"
#include "dkd_hip.h"

namespace {

constexpr int WA_WAVES = 4;
constexpr int N = 49, NP = 49;  // window 7x7; odd LDS pitch
constexpr float NEG = -3.0e38f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const dkd_bf16x4 v = *reinterpret_cast<const dkd_bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
    const dkd_bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<dkd_bf16x4*>(p) = o;
}
// R16: the value a 16-bit tensor holds (policy kernels); the identity for the fp32 kernels
template <bool R16> __device__ __forceinline__ float rq(float v) { return R16 ? rnd16(v) : v; }
template <bool R16> __device__ __forceinline__ f32x4 rq(f32x4 v) {
    if (R16) { v[0] = rnd16(v[0]); v[1] = rnd16(v[1]); v[2] = rnd16(v[2]); v[3] = rnd16(v[3]); }
    return v;
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ f32x4 mfma4(const float4 a, const float4 b, f32x4 c) {  // the four permuted k-steps of one 16-k block
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}
// B operand taken from an accumulator tile (its register r is k-step r)
__device__ __forceinline__ f32x4 mfma4_acc(const float4 a, const f32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ float xor_max(float v, int o) { return fmaxf(v, __shfl_xor(v, o, 64)); }
__device__ __forceinline__ float xor_add(float v, int o) { return v + __shfl_xor(v, o, 64); }

// row fragments: f[t][kb] = X[min(16t + c, N-1)][16kb + 4g .. +3]   (rows past the window shadow the last row)
template <int KB, typename T>
__device__ __forceinline__ void load_rows(float4 (&f)[4][KB], const T* __restrict__ base, long rs, int g, int c, float mul) {
    constexpr bool R16 = dkd_is16<T>::value;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = min(16 * t + c, N - 1);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float4 v = ld4(base + (long)row * rs + 16 * kb + 4 * g);
            f[t][kb] = make_float4(rq<R16>(v.x * mul), rq<R16>(v.y * mul), rq<R16>(v.z * mul), rq<R16>(v.w * mul));
        }
    }
}
// column fragment: X[min(16t + 4g + s, N-1)][16dt + c] for s = 0..3 (the "transposed" operand of the second products)
template <typename T>
__device__ __forceinline__ float4 load_col(const T* __restrict__ base, long rs, int t, int dt, int g, int c) {
    const int r0 = 16 * t + 4 * g;
    const T* p = base + 16 * dt + c;
    return make_float4((float)p[(long)min(r0, N - 1) * rs], (float)p[(long)min(r0 + 1, N - 1) * rs],
                       (float)p[(long)min(r0 + 2, N - 1) * rs], (float)p[(long)min(r0 + 3, N - 1) * rs]);
}

// logits of the T layout from accumulators: + bias + shift mask, padded keys -> NEG; returns per-it column max / fills p
template <int KB, bool R16>
__device__ __forceinline__ void logits_T(f32x4 (&st)[4][4], const float4 (&kf)[4][KB], const float4 (&qf)[4][KB],
                                         const float* __restrict__ sBias, const int* __restrict__ labp, int g, int c) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) acc = mfma4(kf[jt][kb], qf[it][kb], acc);
            st[jt][it] = rq<R16>(acc);   // (policy: q k^T is a 16-bit matmul result; the bias add below is fp32)
        }
    int lab_i[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) lab_i[it] = labp ? labp[min(16 * it + c, N - 1)] : 0;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            const int lab_j = labp ? labp[min(j, N - 1)] : 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = 16 * it + c;
                float v = st[jt][it][r];
                if (j < N) {
                    if (i < N) v += sBias[i * NP + j];
                    if (labp && lab_j != lab_i[it]) v -= 100.0f;
                } else {
                    v = NEG;
                }
                st[jt][it][r] = v;
            }
        }
}

// softmax over the keys of the T layout, in place: st <- exp(st - m_i); returns 1 / l_i per it
__device__ __forceinline__ void softmax_T(f32x4 (&st)[4][4], float (&inv)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        float m = NEG;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, st[jt][it][r]);
        m = xor_max(xor_max(m, 16), 32);
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(st[jt][it][r] - m);
                st[jt][it][r] = p;
                l += p;
            }
        l = xor_add(xor_add(l, 16), 32);
        inv[it] = 1.0f / l;
    }
}

template <int HD, typename T>
__global__ __launch_bounds__(WA_WAVES * 64) void winattn_mfma_fwd_kernel(dkd_winattn_args a) {
    constexpr int KB = HD / 16, DT = HD / 16;
    constexpr bool R16 = dkd_is16<T>::value;
    __shared__ float sBias[N * NP];
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) sBias[e] = a.bias[(long)e * a.H + h];
    __syncthreads();
    const T* aq = reinterpret_cast<const T*>(a.q);
    const T* ak = reinterpret_cast<const T*>(a.k);
    const T* av = reinterpret_cast<const T*>(a.v);
    T* ao = reinterpret_cast<T*>(a.o);
    for (int w = blockIdx.x * WA_WAVES + wave; w < a.Bw; w += gridDim.x * WA_WAVES) {
        const long row0 = (long)w * N;
        const int* labp = a.labels ? a.labels + (long)(w % a.nW) * N : nullptr;
        f32x4 st[4][4];
        {
            float4 kf[4][KB], qf[4][KB];
            load_rows<KB>(kf, ak + row0 * a.k_rs + h * HD, a.k_rs, g, c, 1.0f);
            load_rows<KB>(qf, aq + row0 * a.q_rs + h * HD, a.q_rs, g, c, a.scale);
            logits_T<KB, R16>(st, kf, qf, sBias, labp, g, c);
        }
        float inv[4];
        softmax_T(st, inv);
        if (R16) {  // the probabilities are cast down for the second matmul
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[jt][it][r] = rnd16(st[jt][it][r] * inv[it]);
#pragma unroll
            for (int it = 0; it < 4; ++it) inv[it] = 1.0f;
        }
        f32x4 ot[DT][4];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int it = 0; it < 4; ++it) ot[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
        const T* vb = av + row0 * a.v_rs + h * HD;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const float4 vf = load_col(vb, a.v_rs, jt, dt, g, c);
#pragma unroll
                for (int it = 0; it < 4; ++it) ot[dt][it] = mfma4_acc(vf, st[jt][it], ot[dt][it]);
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = 16 * it + c;
            if (i >= N) continue;
            T* op = ao + (row0 + i) * a.o_rs + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                st4(op + 16 * dt, make_float4(ot[dt][it][0] * inv[it], ot[dt][it][1] * inv[it], ot[dt][it][2] * inv[it],
                                              ot[dt][it][3] * inv[it]));
        }
    }
}

// 16-lane (same g) reductions for the N layout, where a query row lives in lanes c = 0..15 of one group
__device__ __forceinline__ float row16_max(float v) {
    v = xor_max(v, 1); v = xor_max(v, 2); v = xor_max(v, 4); return xor_max(v, 8);
}
__device__ __forceinline__ float row16_add(float v) {
    v = xor_add(v, 1); v = xor_add(v, 2); v = xor_add(v, 4); return xor_add(v, 8);
}

// PART 0: the T-layout half (dQ, d(bias)); PART 1: the N-layout half (dK, dV); PART 2: both in one launch (the default: splitting
// into two launches measured no faster - the hd = 32 halves still need > 128 VGPRs each, i.e. one wave per SIMD).
template <int HD, int PART, typename T>
__global__ __launch_bounds__(WA_WAVES * 64) void winattn_mfma_bwd_kernel(dkd_winattn_args a) {
    constexpr int KB = HD / 16, DT = HD / 16;
    constexpr bool R16 = dkd_is16<T>::value;
    // policy: q enters every product as the 16-bit tensor q * scale, and d(q) = rnd(d(q * scale) * scale)
    const float qmul = R16 ? a.scale : 1.0f, omul = R16 ? 1.0f : a.scale;
    __shared__ float sBias[N * NP];
    // d(bias): ONE tile PER WAVE.  Inside a wave every lane owns its (i, j) of the current sub-tile, so plain read-modify-write
    // accumulates in program order; the four wave tiles are added in a fixed order at the end.  (One shared tile with LDS
    // float atomics made the sum depend on how the waves interleave: gradients were not bit-reproducible run to run.)
    __shared__ float sDbAll[PART != 1 ? WA_WAVES * N * NP : 1];
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        sBias[e] = a.bias[(long)e * a.H + h];
    }
    if (PART != 1)
        for (int e = threadIdx.x; e < WA_WAVES * N * NP; e += blockDim.x) sDbAll[e] = 0.f;
    float* sDb = sDbAll + (PART != 1 ? wave * N * NP : 0);
    __syncthreads();
    for (int w = blockIdx.x * WA_WAVES + wave; w < a.Bw; w += gridDim.x * WA_WAVES) {
        const long row0 = (long)w * N;
        const int* labp = a.labels ? a.labels + (long)(w % a.nW) * N : nullptr;
        const T* qb = reinterpret_cast<const T*>(a.q) + row0 * a.q_rs + h * HD;
        const T* kb_ = reinterpret_cast<const T*>(a.k) + row0 * a.k_rs + h * HD;
        const T* vb = reinterpret_cast<const T*>(a.v) + row0 * a.v_rs + h * HD;
        const T* gb = reinterpret_cast<const T*>(a.d_o) + row0 * a.do_rs + h * HD;
        // ================= T layout (rows = key j, cols = query i): P^T, dP^T -> dS^T -> dQ, d(bias) =================
        if (PART != 1) {
            f32x4 pt[4][4];
            float inv[4];
            {
                float4 kf[4][KB], qf[4][KB];
                load_rows<KB>(kf, kb_, a.k_rs, g, c, 1.0f);
                load_rows<KB>(qf, qb, a.q_rs, g, c, a.scale);
                logits_T<KB, R16>(pt, kf, qf, sBias, labp, g, c);
            }
            softmax_T(pt, inv);
            f32x4 ds[4][4];
            {
                float4 vf[4][KB], gf[4][KB];
                load_rows<KB>(vf, vb, a.v_rs, g, c, 1.0f);
                load_rows<KB>(gf, gb, a.do_rs, g, c, 1.0f);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc = mfma4(vf[jt][kb], gf[it][kb], acc);
                        ds[jt][it] = rq<R16>(acc);  // dP^T
                    }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float delta = 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pt[jt][it][r] *= inv[it];
                        delta = fmaf(pt[jt][it][r], ds[jt][it][r], delta);
                    }
                delta = xor_add(xor_add(delta, 16), 32);
                const int i = 16 * it + c;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = pt[jt][it][r] * (ds[jt][it][r] - delta);
                        ds[jt][it][r] = rq<R16>(v);  // dS^T (policy: the logits tensor is 16-bit, the bias gradient fp32)
                        const int j = 16 * jt + 4 * g + r;
                        if (i < N && j < N) sDb[i * NP + j] += v;
                    }
            }
            // dQ^T[d][i] = scale * sum_j K[j][d] dS^T[j][i]
            f32x4 dq[DT][4];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int it = 0; it < 4; ++it) dq[dt][it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float4 kc = load_col(kb_, a.k_rs, jt, dt, g, c);
#pragma unroll
                    for (int it = 0; it < 4; ++it) dq[dt][it] = mfma4_acc(kc, ds[jt][it], dq[dt][it]);
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = 16 * it + c;
                if (i >= N) continue;
                T* p = reinterpret_cast<T*>(a.dq) + (row0 + i) * a.dq_rs + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    st4(p + 16 * dt, make_float4(rq<R16>(dq[dt][it][0]) * a.scale, rq<R16>(dq[dt][it][1]) * a.scale,
                                                 rq<R16>(dq[dt][it][2]) * a.scale, rq<R16>(dq[dt][it][3]) * a.scale));
            }
        }
        // ================= N layout (rows = query i = 16it + 4g + r, cols = key j = 16jt + c): P, dP -> dS -> dK, dV =================
        if (PART != 0) {
            f32x4 pn[4][4];  // [it][jt]
            {
                float4 kf[4][KB], qf[4][KB];
                load_rows<KB>(kf, kb_, a.k_rs, g, c, 1.0f);
                load_rows<KB>(qf, qb, a.q_rs, g, c, a.scale);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc = mfma4(qf[it][kb], kf[jt][kb], acc);
                        pn[it][jt] = rq<R16>(acc);
                    }
            }
            int lab_j[4];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) lab_j[jt] = labp ? labp[min(16 * jt + c, N - 1)] : 0;
            float inv[4][4];  // [it][r]
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + 4 * g + r;
                    const int lab_i = labp ? labp[min(i, N - 1)] : 0;
                    float m = NEG;
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        const int j = 16 * jt + c;
                        float v = pn[it][jt][r];
                        if (j < N) {
                            if (i < N) v += sBias[i * NP + j];
                            if (labp && lab_j[jt] != lab_i) v -= 100.0f;
                        } else {
                            v = NEG;
                        }
                        pn[it][jt][r] = v;
                        m = fmaxf(m, v);
                    }
                    m = row16_max(m);
                    float l = 0.f;
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        const float p = __expf(pn[it][jt][r] - m);
                        pn[it][jt][r] = p;
                        l += p;
                    }
                    l = row16_add(l);
                    inv[it][r] = i < N ? 1.0f / l : 0.f;  // padded queries contribute nothing to dK / dV
                }
            f32x4 ds[4][4];  // dP, then dS  [it][jt]
            {
                float4 vf[4][KB], gf[4][KB];
                load_rows<KB>(vf, vb, a.v_rs, g, c, 1.0f);
                load_rows<KB>(gf, gb, a.do_rs, g, c, 1.0f);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < KB; ++kb) acc = mfma4(gf[it][kb], vf[jt][kb], acc);
                        ds[it][jt] = rq<R16>(acc);
                    }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float delta = 0.f;
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        pn[it][jt][r] *= inv[it][r];
                        delta = fmaf(pn[it][jt][r], ds[it][jt][r], delta);
                    }
                    delta = row16_add(delta);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        ds[it][jt][r] = rq<R16>(pn[it][jt][r] * (ds[it][jt][r] - delta));
                        pn[it][jt][r] = rq<R16>(pn[it][jt][r]);   // (policy: the 16-bit probabilities of the P v matmul)
                    }
                }
            // dK^T[d][j] = scale * sum_i Q[i][d] dS[i][j]   (raw Q columns: the scale is applied at the store)
            // dV^T[d][j] =         sum_i dO[i][d] P[i][j]
            f32x4 dk[DT][4], dv[DT][4];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) dk[dt][jt] = dv[dt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    float4 qc = load_col(qb, a.q_rs, it, dt, g, c);
                    qc = make_float4(rq<R16>(qc.x * qmul), rq<R16>(qc.y * qmul), rq<R16>(qc.z * qmul), rq<R16>(qc.w * qmul));
                    const float4 gc = load_col(gb, a.do_rs, it, dt, g, c);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) {
                        dk[dt][jt] = mfma4_acc(qc, ds[it][jt], dk[dt][jt]);
                        dv[dt][jt] = mfma4_acc(gc, pn[it][jt], dv[dt][jt]);
                    }
                }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int j = 16 * jt + c;
                if (j >= N) continue;
                T* pk = reinterpret_cast<T*>(a.dk) + (row0 + j) * a.dk_rs + h * HD + 4 * g;
                T* pv = reinterpret_cast<T*>(a.dv) + (row0 + j) * a.dv_rs + h * HD + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    st4(pk + 16 * dt, make_float4(dk[dt][jt][0] * omul, dk[dt][jt][1] * omul, dk[dt][jt][2] * omul,
                                                  dk[dt][jt][3] * omul));
                    st4(pv + 16 * dt, make_float4(dv[dt][jt][0], dv[dt][jt][1], dv[dt][jt][2], dv[dt][jt][3]));
                }
            }
        }
    }
    if (PART != 1) {
        __syncthreads();
        float* part = a.dbias_part + ((long)blockIdx.x * a.H + h) * N * N;
        for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
            float t = sDbAll[e];
#pragma unroll
            for (int wv = 1; wv < WA_WAVES; ++wv) t += sDbAll[wv * N * NP + e];
            part[e] = t;
        }
    }
}

}  // namespace

// launched by dkd_winattn_{fwd,bwd}_f32 (swin.hip) after argument checks; N == 49 only
int dkd_winattn_mfma_fwd(const dkd_winattn_args* a, hipStream_t st) {
    dim3 grid(dkd_winattn_parts(a->Bw, a->H), a->H);
    if (a->hd != 32 && a->hd != 16) return DKD_ERR_UNSUPPORTED;
    DKD_IO_TYPE(a->policy16, T,
                if (a->hd == 32) hipLaunchKernelGGL((winattn_mfma_fwd_kernel<32, T>), grid, dim3(WA_WAVES * 64), 0, st, *a);
                else hipLaunchKernelGGL((winattn_mfma_fwd_kernel<16, T>), grid, dim3(WA_WAVES * 64), 0, st, *a));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

int dkd_winattn_mfma_bwd(const dkd_winattn_args* a, hipStream_t st) {
    dim3 grid(dkd_winattn_parts(a->Bw, a->H), a->H);
    if (a->hd != 32 && a->hd != 16) return DKD_ERR_UNSUPPORTED;
    DKD_IO_TYPE(a->policy16, T,
                if (a->hd == 32) hipLaunchKernelGGL((winattn_mfma_bwd_kernel<32, 2, T>), grid, dim3(WA_WAVES * 64), 0, st, *a);
                else hipLaunchKernelGGL((winattn_mfma_bwd_kernel<16, 2, T>), grid, dim3(WA_WAVES * 64), 0, st, *a));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
