// Flash-style multi-head attention in exact fp32 on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), forward and
// backward, for (i) the DINOv2 teacher blocks (head_dim 64, N = P*P+1 tokens; dgrad-only reuse on student tokens,
// train/distillation_module.py:176-177) and (ii) the ScaleKD cross-attention of positional queries over student
// tokens (losses/scalekd.py:299-316; head_dim = D/num_heads = 16/24/32/48...).
//
// Formulation ("everything transposed", so nothing ever crosses lanes or LDS between the two products):
//   S^T = K Q^T     A = K tile from LDS (key on lane), B = Q fragment in registers -> lane (half,q) holds 16 keys
//   softmax over keys: 15 in-lane max/add + one lane^32 exchange (the query sits on the lane)
//   O^T += V^T P^T  A = V tile from LDS (d on lane, key picked per register), B = the S^T accumulator registers AS
//                   THEY ARE (register r of lane (half,q) is P[q][key(r,half)], key(r,h) = (r&3)+8(r>>2)+4h)
// The backward kernels reuse the same trick: dQ^T += K^T dS^T (queries resident, keys streamed) and
// dK^T += Q^T dS, dV^T += dO^T P (keys resident, queries streamed): no atomics, deterministic.
// k-index convention for the 32x32x2 products over head_dim: lane half h covers d in [h*HD/2, (h+1)*HD/2).
#include "common.h"
#include "dkd_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// raw v_exp_f32: the arguments here are logit - running max <= 0 and the results are probabilities, so the library
// exp2f's denormal-range scaffolding (compare, two selects, add, ldexp per call: 5 extra VALU ops next to every MFMA)
// buys nothing; results below 2^-126 flush to 0
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
// RND (autocast teacher): eager attention under torch autocast rounds the logits (a matmul result) and the
// probabilities (the second matmul's operand) to bf16.  Logits live in log2 units here: round the natural-unit value.
__device__ __forceinline__ float rnd16(float v) { return (float)(__bf16)v; }
template <bool RND>
__device__ __forceinline__ float rnd_logit(float s_log2) { return RND ? rnd16(s_log2 * LN2) * LOG2E : s_log2; }
template <bool RND>
__device__ __forceinline__ float rnd_prob(float p) { return RND ? rnd16(p) : p; }

__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int HD>
struct Cfg {
    static constexpr int DCH = (HD + 31) / 32;   // 32-wide output chunks along head_dim
    static constexpr int HDP = DCH * 32;          // padded head_dim (zero filled)
    static constexpr int LDW = HDP + 1;           // odd row stride: column reads are conflict-free
    static constexpr int KH = HD / 2;             // k-steps per lane half
    static_assert(HD % 8 == 0, "head_dim must be a multiple of 8");
};

// Cooperative copy of a [32 x HD] tile (rows row0.., global row stride rs) into LDS [32][LDW]; rows >= nrows -> 0.
template <int HD, int NT>
__device__ __forceinline__ void stage_tile(float* __restrict__ S, const float* __restrict__ base, long rs, int row0,
                                           int nrows, int tid) {
    constexpr int LDW = Cfg<HD>::LDW, V4 = HD / 4;
    for (int idx = tid; idx < 32 * V4; idx += NT) {
        const int r = idx / V4, c4 = idx % V4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < nrows) x = *reinterpret_cast<const float4*>(base + (long)(row0 + r) * rs + 4 * c4);
        float* d = S + r * LDW + 4 * c4;
        d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
}
template <int HD, int NT>
__device__ __forceinline__ void zero_pad_cols(float* __restrict__ S, int tid) {
    constexpr int LDW = Cfg<HD>::LDW, HDP = Cfg<HD>::HDP;
    if (HDP > HD)
        for (int idx = tid; idx < 32 * (HDP - HD); idx += NT) S[(idx / (HDP - HD)) * LDW + HD + idx % (HDP - HD)] = 0.f;
}

// load this lane's B-operand fragment of row `row` (HD/2 contiguous floats starting at half*HD/2), scaled
template <int HD>
__device__ __forceinline__ void load_frag(float (&f)[HD / 2], const float* __restrict__ base, long rs, int row, int nrows,
                                          int half, float scale) {
    constexpr int KH = HD / 2;
    if (row < nrows) {
        const float4* p = reinterpret_cast<const float4*>(base + (long)row * rs + half * KH);
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) {
            const float4 x = p[i];
            f[4 * i + 0] = x.x * scale; f[4 * i + 1] = x.y * scale; f[4 * i + 2] = x.z * scale; f[4 * i + 3] = x.w * scale;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KH; ++i) f[i] = 0.f;
    }
}

// acc[r] (lane (half, j)) = sum_d T[lane&31][d] * frag_j[d]   : "column" product, T tile in LDS, frag in registers
template <int HD>
__device__ __forceinline__ f32x16 tile_times_frag(const float* __restrict__ T, const float (&f)[HD / 2], int l31, int half) {
    constexpr int LDW = Cfg<HD>::LDW, KH = HD / 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* row = T + l31 * LDW + half * KH;
#pragma unroll
    for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(row[s], f[s], acc, 0, 0, 0);
    return acc;
}

// out[c] += T^T(chunk c) * P : A = T[frag_row(r,half)][32c + lane&31], B = p[r]
template <int HD>
__device__ __forceinline__ void tileT_times_regs(f32x16 (&out)[Cfg<HD>::DCH], const float* __restrict__ T,
                                                 const f32x16& p, int l31, int half) {
    constexpr int LDW = Cfg<HD>::LDW, DCH = Cfg<HD>::DCH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* row = T + frag_row(r, half) * LDW + l31;
#pragma unroll
        for (int c = 0; c < DCH; ++c) out[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32 * c], p[r], out[c], 0, 0, 0);
    }
}

// store a transposed accumulator (rows = d, cols = token on lane) to X[token][head_off + d]
template <int HD>
__device__ __forceinline__ void store_T(const f32x16 (&acc)[Cfg<HD>::DCH], float* __restrict__ base, long rs, int tok,
                                        int ntok, int half, float mul) {
    if (tok >= ntok) return;
    float* row = base + (long)tok * rs;
#pragma unroll
    for (int c = 0; c < Cfg<HD>::DCH; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * c + 8 * g + 4 * half;
            if (d < HD)
                *reinterpret_cast<float4*>(row + d) = make_float4(acc[c][4 * g] * mul, acc[c][4 * g + 1] * mul,
                                                                  acc[c][4 * g + 2] * mul, acc[c][4 * g + 3] * mul);
        }
}

// ---------------------------------------------------------------------------------------------------------
// K/V tile staging split in two halves so the global loads of tile t+1 fly under the MFMA block of tile t
// (issue early / write late), with the LDS image double-buffered: one barrier per key tile.
template <int HD, int NT>
struct TileRegs {
    static constexpr int V4 = HD / 4, TOTAL = 32 * V4, NV = (TOTAL + NT - 1) / NT;
    static constexpr bool FULL = TOTAL % NT == 0;  // every thread owns NV float4 of the tile: no predicate
    float4 v[NV];
    // Rows past the last key are read from the last key instead of zero-filled (their logits are masked / their
    // probabilities are 0, any finite value does): the loads need no predicate and address tile base (wave-uniform)
    // + a 32-bit per-thread offset.
    __device__ __forceinline__ void load(const float* __restrict__ base, long rs, int row0, int nrows, int tid) {
        const float* tb = base + (long)row0 * rs;
        const int lim = nrows - 1 - row0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT, r = idx / V4, c4 = idx % V4;
            if (FULL || idx < TOTAL)
                v[i] = *reinterpret_cast<const float4*>(tb + (unsigned)(min(r, lim) * (int)rs + 4 * c4));
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
        constexpr int LDW = Cfg<HD>::LDW;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * NT, r = idx / V4, c4 = idx % V4;
            if (FULL || idx < TOTAL) {
                float* d = S + r * LDW + 4 * c4;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        }
    }
};

// The <= REM_MAX queries left over after the last full 32-query wave (257 teacher tokens: 1 such query per head instead of
// a 9th MFMA wave): ONE extra workgroup of the forward launch handles them, all its threads cooperating on one row at a
// time - threads over keys for the logits, (key-group, d) for the output - so it overlaps the MFMA workgroups.
// scratch: prob[Nk] + red[NT] + part[NT] floats of LDS (host guarantees Nk + 2*NT <= 4*TSZ).
template <int HD, int NT>
__device__ void attn_fwd_row(const dkd_attn_args& a, int q, int h, int b, float* __restrict__ scratch, int tid) {
    float* prob = scratch;
    float* red = scratch + ((a.Nk + 3) & ~3);
    const float* qrow = a.q + b * a.q_bs + (long)q * a.q_rs + h * HD;
    const float* Kb = a.k + b * a.k_bs + h * HD;
    const float* Vb = a.v + b * a.v_bs + h * HD;
    float qv[HD];
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 x = *reinterpret_cast<const float4*>(qrow + d);
        qv[d] = x.x; qv[d + 1] = x.y; qv[d + 2] = x.z; qv[d + 3] = x.w;
    }
    const float sc = a.scale * LOG2E;
    float mx = -INFINITY;
    for (int k = tid; k < a.Nk; k += NT) {
        const float* kr = Kb + (long)k * a.k_rs;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 x = *reinterpret_cast<const float4*>(kr + d);
            s += (qv[d] * x.x + qv[d + 1] * x.y) + (qv[d + 2] * x.z + qv[d + 3] * x.w);
        }
        s *= sc;
        prob[k] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float sum = 0.f;
    for (int k = tid; k < a.Nk; k += NT) {
        const float p = exp2f(prob[k] - mx);
        prob[k] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = red[0];
    for (int w = 1; w < NT / 64; ++w) sum += red[w];
    // output: thread = (key group g, d); HD <= 64 uses NT/64 key groups of 64 lanes, wider heads loop over d
    constexpr int DL = HD < 64 ? HD : 64;           // lanes along d
    constexpr int NG = NT / DL;                      // key groups
    float* part = red + NT;
    for (int d0 = 0; d0 < HD; d0 += DL) {
        const int d = d0 + tid % DL, g = tid / DL;
        float acc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u] = 0.f;
        int k = g;
        for (; k + 15 * NG < a.Nk; k += 16 * NG) {  // 16 independent loads in flight per thread
            float vv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) vv[u] = Vb[(long)(k + u * NG) * a.v_rs + d];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = fmaf(prob[k + u * NG], vv[u], acc[u]);
        }
        for (; k < a.Nk; k += NG) acc[0] = fmaf(prob[k], Vb[(long)k * a.v_rs + d], acc[0]);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
        for (int u = 0; u < 16; u += 4) { acc0 += acc[u]; acc1 += acc[u + 1]; acc2 += acc[u + 2]; acc3 += acc[u + 3]; }
        __syncthreads();
        part[tid] = (acc0 + acc1) + (acc2 + acc3);
        __syncthreads();
        if (tid < DL) {
            float o = 0.f;
            for (int gg = 0; gg < NG; ++gg) o += part[gg * DL + tid];
            a.o[b * a.o_bs + (long)q * a.o_rs + h * HD + d0 + tid] = o / sum;
        }
    }
    if (a.lse && tid == 0) a.lse[((long)b * a.H + h) * a.Nq + q] = mx + log2f(sum);
    __syncthreads();
}

// One extra key handled on the VALU (online-softmax update with a single logit): used for the <= REM_MAX keys left over
// after the last full 32-key tile (the CLS token makes the teacher's N = 32 k + 1), instead of a whole padded MFMA tile.
constexpr int REM_MAX = 4;
template <int HD, bool RND = false>
__device__ __forceinline__ void attn_one_key(const float* __restrict__ krow, const float* __restrict__ vrow,
                                             const float (&qf)[HD / 2], f32x16 (&o)[Cfg<HD>::DCH], float& m, float& l, int half) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD / 2; ++d) s = fmaf(krow[half * (HD / 2) + d], qf[d], s);
    s += __shfl_xor(s, 32, 64);
    s = rnd_logit<RND>(s);
    const float mn = fmaxf(m, s);
    const float alpha = exp2_raw(m - mn), p = rnd_prob<RND>(exp2_raw(s - mn));
    l = l * alpha + (half == 0 ? p : 0.f);  // the two halves' partial sums are added at the end: count p once
    m = mn;
#pragma unroll
    for (int c = 0; c < Cfg<HD>::DCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = fmaf(p, vrow[32 * c + frag_row(r, half)], o[c][r] * alpha);
}

template <int HD, int NW, bool RND = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(dkd_attn_args a, int nq_limit) {
    using C = Cfg<HD>;
    constexpr int NT = NW * 64, TSZ = 32 * C::LDW;
    __shared__ float smem[4 * TSZ];  // [buffer][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * NW + wid) * 32;
    const int q = q0 + l31;
    const bool active = q0 < nq_limit;  // wave-uniform: waves past the last query only help staging
    const float* Qb = a.q + b * a.q_bs + h * HD;
    const float* Kb = a.k + b * a.k_bs + h * HD;
    const float* Vb = a.v + b * a.v_bs + h * HD;

    float qf[HD / 2];
    load_frag<HD>(qf, Qb, a.q_rs, q, a.Nq, half, a.scale * LOG2E);
    f32x16 o[C::DCH];
#pragma unroll
    for (int c = 0; c < C::DCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) zero_pad_cols<HD, NT>(smem + t * TSZ, tid);

    // ONE leftover query (the CLS token: N = 32 k + 1) rides along in the key loop of the last workgroup of its (head,
    // image) on the VALU, off the K/V tiles already in LDS: each of the 4 waves takes 8 keys of the tile - lane (j, c) the
    // dims chunk c of key j for the logit, lane d for the output - and keeps its own online-softmax partial; the four
    // partials are merged after the loop.  (Appending the row after the loop re-read K and V from global: a 20 us serial
    // tail on the teacher shape.)
    constexpr bool XLOOP = NW == 4 && HD <= 64;
    constexpr int DPL = HD / 8;
    const bool xrow = XLOOP && a.Nq - nq_limit == 1 && blockIdx.x == gridDim.x - 1;
    float xq[DPL], xm = -INFINITY, xl = 0.f, xo = 0.f;
    if (xrow) {
        const float* qx = Qb + (long)nq_limit * a.q_rs + (lane & 7) * DPL;
#pragma unroll
        for (int i = 0; i < DPL; ++i) xq[i] = qx[i] * (a.scale * LOG2E);
    }

    TileRegs<HD, NT> kr, vr;
    const int nkt = (a.Nk + 31) / 32;
    const int rem = a.Nk & 31;
    const bool valu_tail = rem > 0 && rem <= REM_MAX && nkt > 1;
    kr.load(Kb, a.k_rs, 0, a.Nk, tid);
    vr.load(Vb, a.v_rs, 0, a.Nk, tid);
    kr.store(smem, tid);
    vr.store(smem + TSZ, tid);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const float* Ks = smem + (kt & 1) * 2 * TSZ;
        const float* Vs = Ks + TSZ;
        if (kt + 1 < nkt) {
            kr.load(Kb, a.k_rs, (kt + 1) * 32, a.Nk, tid);
            vr.load(Vb, a.v_rs, (kt + 1) * 32, a.Nk, tid);
        }
        if (xrow) {
            const int j = wid * 8 + (lane >> 3);
            const float* kp = Ks + j * C::LDW + (lane & 7) * DPL;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) s = fmaf(kp[i], xq[i], s);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s = rnd_logit<RND>(s);
            if (kt * 32 + j >= a.Nk) s = -INFINITY;
            float mt = fmaxf(s, __shfl_xor(s, 8, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(xm, mt);
            if (mn > -INFINITY) {  // wave-uniform
                const float alpha = exp2_raw(xm - mn), p = rnd_prob<RND>(exp2_raw(s - mn));
                xl = xl * alpha + p;
                xm = mn;
                const float* vp = Vs + wid * 8 * C::LDW + (lane < C::HDP ? lane : C::HDP - 1);
                float acc = xo * alpha;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 8 * jj)), vp[jj * C::LDW], acc);
                xo = acc;
            }
        }
        if (active && valu_tail && kt == nkt - 1) {
            for (int j = 0; j < rem; ++j) attn_one_key<HD, RND>(Ks + j * C::LDW, Vs + j * C::LDW, qf, o, m, l, half);
        } else if (active) {
            f32x16 s = tile_times_frag<HD>(Ks, qf, l31, half);
            if (RND) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = rnd_logit<true>(s[r]);
            }
            if (kt == nkt - 1 && rem != 0) {  // only the last key tile can hold padded keys
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + frag_row(r, half) >= a.Nk) s[r] = -INFINITY;
            }
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(m, mt);
            const float alpha = exp2_raw(m - mn);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = rnd_prob<RND>(exp2_raw(s[r] - mn));
                ls += s[r];
            }
            l = l * alpha + ls;
            m = mn;
#pragma unroll
            for (int c = 0; c < C::DCH; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
            tileT_times_regs<HD>(o, Vs, s, l31, half);
        }
        if (kt + 1 < nkt) {
            float* Kn = smem + ((kt + 1) & 1) * 2 * TSZ;
            kr.store(Kn, tid);
            vr.store(Kn + TSZ, tid);
        }
        __syncthreads();
    }
    if (active) {
        l += __shfl_xor(l, 32, 64);
        store_T<HD>(o, a.o + b * a.o_bs + h * HD, a.o_rs, q, nq_limit, half, 1.f / l);
        if (a.lse && half == 0 && q < nq_limit) a.lse[((long)b * a.H + h) * a.Nq + q] = m + log2f(l);  // log2 units
    }
    // leftover queries (nq_limit < Nq): the LAST workgroup of each (head, image) appends them after its own tiles.  A
    // separate workgroup for them measured +25 % on the teacher shape: it perturbs the even 3-per-CU placement of the MFMA
    // workgroups (gpurun_out notes in DESIGN.md).
    if (xrow) {
        float lw = wave_sum((lane & 7) == 0 ? xl : 0.f);  // every lane of a key's group carries the same p
        float* sc = smem + wid * 66;
        if (lane < HD) sc[lane] = xo;
        if (lane == 0) { sc[64] = xm; sc[65] = lw; }
        __syncthreads();
        if (wid == 0) {
            float M = smem[64];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, smem[w * 66 + 64]);
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = exp2f(smem[w * 66 + 64] - M);
                L = fmaf(f, smem[w * 66 + 65], L);
                if (lane < HD) O = fmaf(f, smem[w * 66 + lane], O);
            }
            if (lane < HD) a.o[b * a.o_bs + (long)nq_limit * a.o_rs + h * HD + lane] = O / L;
            if (a.lse && lane == 0) a.lse[((long)b * a.H + h) * a.Nq + nq_limit] = M + log2f(L);
        }
    } else if (nq_limit < a.Nq && blockIdx.x == gridDim.x - 1) {
        for (int qq = nq_limit; qq < a.Nq; ++qq) attn_fwd_row<HD, NT>(a, qq, h, b, smem, tid);
    }
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
// hd/4 consecutive lanes read the float4s of one (query, head) row of O and dO (dense, coalesced runs; one lane per row
// made every load instruction touch 64 different cache lines), partial sums meet in LDS.  blockDim is a multiple of G.
__global__ void attn_delta_kernel(dkd_attn_args a, int G) {
    __shared__ float part[256];
    const int tid = threadIdx.x, gi = tid / G, c = tid % G;
    const long row = (long)blockIdx.x * (blockDim.x / G) + gi;  // (b * Nq + q) * H + h
    const long total = (long)a.B * a.H * a.Nq;
    float s = 0.f;
    int q = 0, h = 0, b = 0;
    if (row < total) {
        h = (int)(row % a.H);
        q = (int)((row / a.H) % a.Nq);
        b = (int)(row / ((long)a.H * a.Nq));
        const float4 x = *reinterpret_cast<const float4*>(a.o + b * a.o_bs + (long)q * a.o_rs + h * a.hd + 4 * c);
        const float4 y = *reinterpret_cast<const float4*>(a.d_o + b * a.do_bs + (long)q * a.do_rs + h * a.hd + 4 * c);
        s = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    part[tid] = s;
    __syncthreads();
    if (c == 0 && row < total) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += part[tid + j];
        a.delta[((long)b * a.H + h) * a.Nq + q] = t;
    }
}

// dQ: queries resident (one wave = 32 queries), keys/values streamed through LDS.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(dkd_attn_args a) {
    using C = Cfg<HD>;
    constexpr int NT = NW * 64;
    __shared__ float Ks[32 * C::LDW], Vs[32 * C::LDW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = (blockIdx.x * NW + wid) * 32 + l31;
    const float* Kb = a.k + b * a.k_bs + h * HD;
    const float* Vb = a.v + b * a.v_bs + h * HD;
    float qf[HD / 2], dof[HD / 2];
    load_frag<HD>(qf, a.q + b * a.q_bs + h * HD, a.q_rs, q, a.Nq, half, a.scale * LOG2E);
    load_frag<HD>(dof, a.d_o + b * a.do_bs + h * HD, a.do_rs, q, a.Nq, half, 1.f);
    const long sidx = ((long)b * a.H + h) * a.Nq + q;
    const float lse = q < a.Nq ? a.lse[sidx] : 0.f;
    const float dl = q < a.Nq ? a.delta[sidx] : 0.f;
    f32x16 dq[C::DCH];
#pragma unroll
    for (int c = 0; c < C::DCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[c][r] = 0.f;
    zero_pad_cols<HD, NT>(Ks, tid);
    zero_pad_cols<HD, NT>(Vs, tid);
    const int nkt = (a.Nk + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        stage_tile<HD, NT>(Ks, Kb, a.k_rs, kt * 32, a.Nk, tid);
        stage_tile<HD, NT>(Vs, Vb, a.v_rs, kt * 32, a.Nk, tid);
        __syncthreads();
        f32x16 s = tile_times_frag<HD>(Ks, qf, l31, half);    // S^T (log2 units)
        f32x16 dp = tile_times_frag<HD>(Vs, dof, l31, half);  // dP^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool valid = kt * 32 + frag_row(r, half) < a.Nk;
            const float p = valid ? exp2_raw(s[r] - lse) : 0.f;
            s[r] = p * (dp[r] - dl) * a.scale;  // dS^T
        }
        tileT_times_regs<HD>(dq, Ks, s, l31, half);
    }
    store_T<HD>(dq, a.dq + b * a.dq_bs + h * HD, a.dq_rs, q, a.Nq, half, 1.f);
}

// dK, dV: keys resident (one wave = 32 keys), queries / dO streamed through LDS.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(dkd_attn_args a) {
    using C = Cfg<HD>;
    constexpr int NT = NW * 64;
    __shared__ float Qs[32 * C::LDW], Os[32 * C::LDW];
    __shared__ float lse_s[32], dl_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int key = (blockIdx.x * NW + wid) * 32 + l31;
    const float* Qb = a.q + b * a.q_bs + h * HD;
    const float* Ob = a.d_o + b * a.do_bs + h * HD;
    float kf[HD / 2], vf[HD / 2];
    load_frag<HD>(kf, a.k + b * a.k_bs + h * HD, a.k_rs, key, a.Nk, half, 1.f);
    load_frag<HD>(vf, a.v + b * a.v_bs + h * HD, a.v_rs, key, a.Nk, half, 1.f);
    f32x16 dk[C::DCH], dv[C::DCH];
#pragma unroll
    for (int c = 0; c < C::DCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[c][r] = dv[c][r] = 0.f;
    zero_pad_cols<HD, NT>(Qs, tid);
    zero_pad_cols<HD, NT>(Os, tid);
    const float sc2 = a.scale * LOG2E;
    const long sbase = ((long)b * a.H + h) * a.Nq;
    const int nqt = (a.Nq + 31) / 32;
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
        stage_tile<HD, NT>(Qs, Qb, a.q_rs, qt * 32, a.Nq, tid);
        stage_tile<HD, NT>(Os, Ob, a.do_rs, qt * 32, a.Nq, tid);
        if (tid < 32) {
            const int qq = qt * 32 + tid;
            lse_s[tid] = qq < a.Nq ? a.lse[sbase + qq] : INFINITY;  // +inf -> p = 0 for padded queries
            dl_s[tid] = qq < a.Nq ? a.delta[sbase + qq] : 0.f;
        }
        __syncthreads();
        f32x16 s = tile_times_frag<HD>(Qs, kf, l31, half);   // S[q(r,half)][key]
        f32x16 dp = tile_times_frag<HD>(Os, vf, l31, half);  // dP[q][key]
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = frag_row(r, half);
            const float p = exp2_raw(s[r] * sc2 - lse_s[qr]);
            s[r] = p;
            ds[r] = p * (dp[r] - dl_s[qr]) * a.scale;
        }
        tileT_times_regs<HD>(dv, Os, s, l31, half);
        tileT_times_regs<HD>(dk, Qs, ds, l31, half);
    }
    store_T<HD>(dk, a.dk + b * a.dk_bs + h * HD, a.dk_rs, key, a.Nk, half, 1.f);
    store_T<HD>(dv, a.dv + b * a.dv_bs + h * HD, a.dv_rs, key, a.Nk, half, 1.f);
}

int pick_nw(int n) {  // waves per workgroup (32 rows each): least padded rows, prefer larger
    const int w4 = (n + 127) / 128 * 128 - n, w2 = (n + 63) / 64 * 64 - n;
    return w2 < w4 ? 2 : 4;
}
// forward: waves past the last query skip the matrix work but still occupy a wave slot for the whole key loop, so pick
// the workgroup size in {4, 3, 2} waves that wastes the fewest slots (257 teacher tokens = 9 query waves -> 3 x 3;
// 5-wave workgroups measured 10-60 % slower: one SIMD carries two of the waves)
int pick_nw_fwd(int n) {
    const int nwaves = (n + 31) / 32;
    int best = 4, best_slots = (nwaves + 3) / 4 * 4;
    for (int nw = 3; nw >= 2; --nw) {
        const int slots = (nwaves + nw - 1) / nw * nw;
        if (slots < best_slots) { best = nw; best_slots = slots; }
    }
    return best;
}

bool strides_ok(const dkd_attn_args* a, bool bwd) {
    const int64_t s[] = {a->q_bs, a->q_rs, a->k_bs, a->k_rs, a->v_bs, a->v_rs, a->o_bs, a->o_rs};
    for (int64_t x : s)
        if (x & 3) return false;
    if (bwd) {
        const int64_t t[] = {a->do_bs, a->do_rs, a->dq_bs, a->dq_rs, a->dk_bs, a->dk_rs, a->dv_bs, a->dv_rs};
        for (int64_t x : t)
            if (x & 3) return false;
    }
    return true;
}

template <int HD>
int launch_fwd(const dkd_attn_args* a, hipStream_t st) {
    // queries left over after the last full wave go to the row kernel when there are only a few of them
    const int qrem = a->Nq & 31;
    const bool row_tail = qrem > 0 && qrem <= REM_MAX && a->Nq > 32 && a->Nk + 2 * 256 + 4 <= 4 * 32 * Cfg<HD>::LDW;
    const int nq_main = row_tail ? a->Nq - qrem : a->Nq;
    const int nw = pick_nw_fwd(nq_main);
    dim3 grid((nq_main + 32 * nw - 1) / (32 * nw), a->H, a->B);
    if (a->round_bf16) {  // the autocast teacher (DINOv2 heads are 64 wide): always the 4-wave kernel
        if (HD != 64) return DKD_ERR_UNSUPPORTED;
        const dim3 grid4((nq_main + 127) / 128, a->H, a->B);
        hipLaunchKernelGGL((attn_fwd_kernel<HD == 64 ? 64 : HD, 4, HD == 64>), grid4, dim3(256), 0, st, *a, nq_main);
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
    switch (nw) {
        case 1: hipLaunchKernelGGL((attn_fwd_kernel<HD, 1>), grid, dim3(64), 0, st, *a, nq_main); break;
        case 2: hipLaunchKernelGGL((attn_fwd_kernel<HD, 2>), grid, dim3(128), 0, st, *a, nq_main); break;
        case 3: hipLaunchKernelGGL((attn_fwd_kernel<HD, 3>), grid, dim3(192), 0, st, *a, nq_main); break;
        default: hipLaunchKernelGGL((attn_fwd_kernel<HD, 4>), grid, dim3(256), 0, st, *a, nq_main); break;
    }

    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

template <int HD>
int launch_bwd(const dkd_attn_args* a, hipStream_t st) {
    const long total = (long)a->B * a->H * a->Nq;
    const int G = HD / 4, bd = 256 / G * G, rpb = bd / G;  // block: a whole number of (query, head) rows
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + rpb - 1) / rpb)), dim3(bd), 0, st, *a, G);
    DKD_LAUNCH_CHECK();
    int nw = pick_nw(a->Nq);
    dim3 gq((a->Nq + 32 * nw - 1) / (32 * nw), a->H, a->B);
    if (nw == 2) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 2>), gq, dim3(128), 0, st, *a);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4>), gq, dim3(256), 0, st, *a);
    DKD_LAUNCH_CHECK();
    nw = pick_nw(a->Nk);
    dim3 gk((a->Nk + 32 * nw - 1) / (32 * nw), a->H, a->B);
    if (nw == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 2>), gk, dim3(128), 0, st, *a);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4>), gk, dim3(256), 0, st, *a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

}  // namespace

#define DKD_HD_DISPATCH(FN, a, st)                     \
    switch ((a)->hd) {                                 \
        case 8: return FN<8>(a, st);                   \
        case 16: return FN<16>(a, st);                 \
        case 24: return FN<24>(a, st);                 \
        case 32: return FN<32>(a, st);                 \
        case 40: return FN<40>(a, st);                 \
        case 48: return FN<48>(a, st);                 \
        case 64: return FN<64>(a, st);                 \
        case 80: return FN<80>(a, st);                 \
        case 96: return FN<96>(a, st);                 \
        default: return DKD_ERR_UNSUPPORTED;           \
    }

extern "C" int dkd_attn_fwd_f32(const dkd_attn_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->v || !a->o || a->B <= 0 || a->H <= 0 || a->Nq <= 0 || a->Nk <= 0)
        return DKD_ERR_BAD_ARG;
    if (!strides_ok(a, false)) return DKD_ERR_ALIGN;
    DKD_HD_DISPATCH(launch_fwd, a, (hipStream_t)stream);
}

extern "C" int dkd_attn_bwd_f32(const dkd_attn_args* a, void* stream) {
    if (!a || !a->q || !a->k || !a->v || !a->o || !a->d_o || !a->lse || !a->delta || !a->dq || !a->dk || !a->dv ||
        a->B <= 0 || a->H <= 0 || a->Nq <= 0 || a->Nk <= 0)
        return DKD_ERR_BAD_ARG;
    if (!strides_ok(a, true)) return DKD_ERR_ALIGN;
    DKD_HD_DISPATCH(launch_bwd, a, (hipStream_t)stream);
}
