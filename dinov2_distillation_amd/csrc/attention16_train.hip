// The 16-bit policy's attention WITH gradients, flash-style on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), head dims 16 / 24 /
// 32 (tiles padded to 32: HDP = 32), 40 / 48 / 64 (HDP = 64) and 72 / 80 / 96 (HDP = 96: the 16-head projector on 1536 channels, MiT's
// 80-wide third stage): the
// ScaleKD projectors' cross attention (losses/scalekd.py:308-313) and the frozen DINOv2 blocks re-used on student tokens
// (train/distillation_module.py:176-177) under the reference's default precision=16 (train.py:263 == torch autocast).
//
// Same arithmetic as the materialised path (ops.attention16_*: batched bf16 GEMMs around dkd_attn16_softmax_*), i.e. the roundings
// of EAGER attention under autocast and of autograd's backward through it, without the [B*H, Nq, Nk] logit / probability tensors:
//   S = rnd(q k^T); T = rnd(rnd(S c1) c2); P32 = softmax_fp32(T) with the FINAL row max and sum; P = rnd(P32); O = rnd(P v)
//   dP = rnd(dO v^T); dT = P32 (dP - sum_k dP P32); dS = rnd(rnd(rnd(dT) c2) c1); dq = rnd(dS k); dk = rnd(dS^T q); dv = rnd(P^T dO)
// The final max / sum need the whole row before the first probability can be rounded, so the forward sweeps the keys twice
// (statistics, then P v); the backward recomputes S and P32 from q, k and the saved statistics, and sweeps twice as well where the
// row term sum_k dP P32 is needed before dS (the dq kernel, which also leaves it in `delta` for the dk / dv kernel).
//
// Layouts ("everything transposed", attention.hip / attention16_flash.hip): whichever token set is RESIDENT in a wave sits on the
// lanes (B operands held in registers: rows of q / dO for the forward and dq kernels, rows of k / v for the dk / dv kernel), the
// STREAMED set arrives as 64-row tiles in LDS (LDS-DMA, two stages, one barrier per tile) and lands on the accumulator registers
// - which, packed to bf16 eight at a time, are the B operand of the second product, whose A operand is the streamed tile read
// transposed (ds_read_b64_tr_b16).  One LDS image serves both read forms: rows of 2 HDP bytes whose 16-byte chunk index is XOR-ed
// with the row's position in its 256-byte bank period group ((row / 2) % 8 for 128-byte rows, (row / 4) % 4 for 64-byte rows; 192-byte
// rows start 48 banks apart - a period of four rows - and XOR (row / 4) % 4 into the low two bits of the chunk index, which stays inside
// its group of four of the 12 chunks); chunks past the real head dim (24 in a 32-wide tile) are never loaded and stay zero.
#include <cmath>

#include "common.h"
#include "dkd_hip.h"
#include "gemm_bf16_common.h"

namespace {
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr float LOG2E = 1.4426950408889634f;
constexpr int TR = 64, NW = 4;

__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int HDP>
struct Geo {
    static constexpr int ROWB = HDP * 2, TILE = TR * ROWB, CPR = HDP / 8, PIECES = TILE / 1024, PPW = PIECES / NW;
    static constexpr int KC = HDP / 16, DB = HDP / 32;
    static_assert(HDP == 96 || HDP == 64 || HDP == 32, "tiles are 32, 64 or 96 wide");
    static __device__ __forceinline__ int swz(int row) { return HDP == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};

// one wave's PPW 1-KiB pieces of a 64-row tile; rows past `nrows` repeat the last row (finite values that the masks below turn into
// exact zeros); chunks at or past the real head dim are not loaded (their LDS slots were zeroed once and are never written)
template <int HDP>
__device__ __forceinline__ void issue_tile(const __bf16* __restrict__ base, long ld, int row0, int nrows, int hd, char* __restrict__ S, int wid,
                                           int lane) {
    using G = Geo<HDP>;
#pragma unroll
    for (int p = 0; p < G::PPW; ++p) {
        // the piece's lane i lands at LDS byte 1024 piece + 16 i = row (64 piece + i) / CPR, slot (64 piece + i) % CPR
        const int piece = wid + NW * p, at = 64 * piece + lane, row = at / G::CPR, chunk = (at % G::CPR) ^ G::swz(row);
        const __bf16* src = base + (long)min(row0 + row, nrows - 1) * ld + 8 * chunk;
        if (8 * chunk < hd) __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(S + piece * 1024), 16, 0, 0);
    }
}
template <int BYTES>
__device__ __forceinline__ void zero_lds(char* __restrict__ S, int tid) {
    for (int i = tid; i < BYTES / 16; i += NW * 64) reinterpret_cast<uint4*>(S)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
}
// A operand, rows r0 + (lane & 31) of a tile, k = d in [16 kc + 8 half, +8)
template <int HDP>
__device__ __forceinline__ bf16x8 row_frag(const char* __restrict__ T, int r0, int kc, int lane) {
    using G = Geo<HDP>;
    const int row = r0 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(T + row * G::ROWB + 16 * ((2 * kc + (lane >> 5)) ^ G::swz(row)));
}
// A operand of a TRANSPOSED tile: rows = d in [32 c + (lane & 31)], k = tile rows rb + 4 half + {0..3} and rb + 8 + 4 half + {0..3}
// (the token order the accumulator registers 8 b .. 8 b + 7 of the other product hold, rb = 32 block + 16 b)
template <int HDP>
__device__ __forceinline__ bf16x8 tr_frag(const char* __restrict__ T, int rb, int c, int lane) {
    using G = Geo<HDP>;
    const int half = lane >> 5, cb = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
    const int r_lo = rb + 4 * half + qq, r_hi = r_lo + 8, ch = 4 * c + 2 * cb + (pp >> 1);
    union { s16x4 v[2]; bf16x8 f; } u;
    u.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + r_lo * G::ROWB + 16 * (ch ^ G::swz(r_lo)) + 8 * (pp & 1)));
    u.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(T + r_hi * G::ROWB + 16 * (ch ^ G::swz(r_hi)) + 8 * (pp & 1)));
    return u.f;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int b16) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[8 * b16 + e];
    return o;
}
// this lane's resident row (token `row` of a [rows, ld] slice at the column offset of the head): KC fragments of 8 d, zero past hd
template <int HDP>
__device__ __forceinline__ void load_resident(bf16x8 (&f)[Geo<HDP>::KC], const __bf16* __restrict__ base, long ld, int row, int nrows,
                                              int hd, int half) {
    const __bf16* p = base + (long)min(row, nrows - 1) * ld + 8 * half;
#pragma unroll
    for (int kc = 0; kc < Geo<HDP>::KC; ++kc) {
        if (16 * kc + 8 * half < hd) f[kc] = *reinterpret_cast<const bf16x8*>(p + 16 * kc);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) f[kc][e] = (__bf16)0.f;
    }
}
// the 16-bit logit chain in log2 units, T = rnd(rnd(rnd(acc) c1) c2), and its backward dS = rnd(rnd(rnd(dT) c2) c1).  The kernels are
// bound by this per-element VALU work (three evaluations per logit in the backward pass), so the roundings that are exact are not
// executed: MODE 1: c1 is a power of two (head_dim 64: 2^-3) - rnd(x) c1 needs no second rounding; MODE 2: additionally c2 == 1
// (the re-used teacher blocks).  k12 = c1 c2 LOG2E, c12 = c1 c2.
struct Scales { float c1, c2, c12, k12; };
template <int MODE>
__device__ __forceinline__ float logit2(float acc, const Scales& s) {
    if (MODE == 2) return rnd16(acc) * s.k12;
    if (MODE == 1) return rnd16(rnd16(acc) * s.c12) * LOG2E;
    return rnd16(rnd16(rnd16(acc) * s.c1) * s.c2) * LOG2E;
}
template <int MODE>
__device__ __forceinline__ float dlogit(float dT, const Scales& s) {
    if (MODE == 2) return rnd16(dT) * s.c1;
    if (MODE == 1) return rnd16(rnd16(dT) * s.c2) * s.c1;
    return rnd16(rnd16(rnd16(dT) * s.c2) * s.c1);
}

// S^T block (32 streamed rows x this lane's resident token): acc[r] <-> streamed row r0 + frag_row(r, half)
template <int HDP>
__device__ __forceinline__ f32x16 product_T(const char* __restrict__ T, int r0, const bf16x8 (&res)[Geo<HDP>::KC], int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < Geo<HDP>::KC; ++kc)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HDP>(T, r0, kc, lane), res[kc], acc, 0, 0, 0);
    return acc;
}
// out^T (HDP d x resident token) += tile^T x packed accumulators of one 32-row block
template <int HDP>
__device__ __forceinline__ void accumulate_T(f32x16 (&out)[Geo<HDP>::DB], const char* __restrict__ T, int r0, const f32x16& p, int lane) {
#pragma unroll
    for (int b16 = 0; b16 < 2; ++b16) {
        const bf16x8 pf = pack8(p, b16);
#pragma unroll
        for (int c = 0; c < Geo<HDP>::DB; ++c)
            out[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HDP>(T, r0 + 16 * b16, c, lane), pf, out[c], 0, 0, 0);
    }
}
// store a transposed accumulator set (rows = d, lane = token) as the token's hd bf16 values
template <int HDP>
__device__ __forceinline__ void store_T(const f32x16 (&acc)[Geo<HDP>::DB], __bf16* __restrict__ row, int hd, int half) {
#pragma unroll
    for (int c = 0; c < Geo<HDP>::DB; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * c + 8 * g + 4 * half;
            if (d < hd) {
                const bf16x4 v = {(__bf16)acc[c][4 * g], (__bf16)acc[c][4 * g + 1], (__bf16)acc[c][4 * g + 2], (__bf16)acc[c][4 * g + 3]};
                *reinterpret_cast<bf16x4*>(row + d) = v;
            }
        }
}

struct Ptrs {
    const __bf16 *q, *k, *v, *d_o;
    __bf16 *o, *dq, *dk, *dv;
    float *m, *linv, *delta;
    long ldq, ldkv, ldo, lddo, lddq, lddkv;
    long q_bs;   // q batch stride in elements (0: one query block for every image)
    int B, H, Nq, Nk, hd;
    float c1, c2;
};

// ---- forward: queries resident; sweep 1 = row max / sum over all keys, sweep 2 = P v ------------------------------------------
template <int HDP, int MODE>
__global__ __launch_bounds__(NW * 64) void attn16_train_fwd_kernel(Ptrs a) {
    using G = Geo<HDP>;
    constexpr int TILE = G::TILE;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE];  // [stage][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q = (blockIdx.x * NW + wid) * 32 + l31;
    const __bf16* Kb = a.k + (long)b * a.Nk * a.ldkv + h * a.hd;
    const __bf16* Vb = a.v + (long)b * a.Nk * a.ldkv + h * a.hd;
    bf16x8 qf[G::KC];
    zero_lds<4 * TILE>(smem, tid);
    load_resident<HDP>(qf, a.q + b * a.q_bs + h * a.hd, a.ldq, q, a.Nq, a.hd, half);
    const int nkt = (a.Nk + TR - 1) / TR;
    const Scales sc{a.c1, a.c2, a.c1 * a.c2, a.c1 * a.c2 * LOG2E};
    float m = -INFINITY, l = 0.f;
    // sweep 1
    issue_tile<HDP>(Kb, a.ldkv, 0, a.Nk, a.hd, smem, wid, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        wait_dma();
        __builtin_amdgcn_s_barrier();
        const char* Ks = smem + (kt & 1) * 2 * TILE;
        if (kt + 1 < nkt) issue_tile<HDP>(Kb, a.ldkv, (kt + 1) * TR, a.Nk, a.hd, smem + ((kt + 1) & 1) * 2 * TILE, wid, lane);
        f32x16 s[2];
        float mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            s[kb] = product_T<HDP>(Ks, 32 * kb, qf, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float u = logit2<MODE>(s[kb][r], sc);
                if (kt == nkt - 1 && kt * TR + 32 * kb + frag_row(r, half) >= a.Nk) u = -INFINITY;   // (only the last tile is ragged)
                s[kb][r] = u;
                mt = fmaxf(mt, u);
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ls += exp2_raw(s[kb][r] - mn);
        l = l * exp2_raw(m - mn) + ls;
        m = mn;
    }
    l += __shfl_xor(l, 32, 64);
    const float linv = 1.0f / l;
    // sweep 2 (the barrier at its first tile also fences the last reads of sweep 1)
    f32x16 o[G::DB];
#pragma unroll
    for (int c = 0; c < G::DB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    __builtin_amdgcn_s_barrier();
    issue_tile<HDP>(Kb, a.ldkv, 0, a.Nk, a.hd, smem, wid, lane);
    issue_tile<HDP>(Vb, a.ldkv, 0, a.Nk, a.hd, smem + TILE, wid, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        wait_dma();
        __builtin_amdgcn_s_barrier();
        const char* Ks = smem + (kt & 1) * 2 * TILE;
        const char* Vs = Ks + TILE;
        if (kt + 1 < nkt) {
            char* nx = smem + ((kt + 1) & 1) * 2 * TILE;
            issue_tile<HDP>(Kb, a.ldkv, (kt + 1) * TR, a.Nk, a.hd, nx, wid, lane);
            issue_tile<HDP>(Vb, a.ldkv, (kt + 1) * TR, a.Nk, a.hd, nx + TILE, wid, lane);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = product_T<HDP>(Ks, 32 * kb, qf, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float u = logit2<MODE>(s[r], sc);
                const bool dead = kt == nkt - 1 && kt * TR + 32 * kb + frag_row(r, half) >= a.Nk;
                s[r] = dead ? 0.f : rnd16(exp2_raw(u - m) * linv);
            }
            accumulate_T<HDP>(o, Vs, 32 * kb, s, lane);
        }
    }
    if (q < a.Nq) {
        store_T<HDP>(o, a.o + ((long)b * a.Nq + q) * a.ldo + h * a.hd, a.hd, half);
        if (half == 0) {
            const long si = ((long)b * a.H + h) * a.Nq + q;
            a.m[si] = m;
            a.linv[si] = linv;
        }
    }
}

// ---- backward, queries resident: delta = sum_k dP P32 (sweep 1), dq = rnd(dS k) (sweep 2) ---------------------------------------
template <int HDP, int MODE>
__global__ __launch_bounds__(NW * 64) void attn16_train_bwd_dq_kernel(Ptrs a) {
    using G = Geo<HDP>;
    constexpr int TILE = G::TILE;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q = (blockIdx.x * NW + wid) * 32 + l31;
    const __bf16* Kb = a.k + (long)b * a.Nk * a.ldkv + h * a.hd;
    const __bf16* Vb = a.v + (long)b * a.Nk * a.ldkv + h * a.hd;
    bf16x8 qf[G::KC], gf[G::KC];
    zero_lds<4 * TILE>(smem, tid);
    load_resident<HDP>(qf, a.q + b * a.q_bs + h * a.hd, a.ldq, q, a.Nq, a.hd, half);
    load_resident<HDP>(gf, a.d_o + (long)b * a.Nq * a.lddo + h * a.hd, a.lddo, q, a.Nq, a.hd, half);
    const long si = ((long)b * a.H + h) * a.Nq + min(q, a.Nq - 1);
    const float m = a.m[si], linv = a.linv[si];
    const int nkt = (a.Nk + TR - 1) / TR;
    const Scales sc{a.c1, a.c2, a.c1 * a.c2, a.c1 * a.c2 * LOG2E};
    float delta = 0.f;
    f32x16 dq[G::DB];
#pragma unroll
    for (int c = 0; c < G::DB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[c][r] = 0.f;
    for (int sweep = 0; sweep < 2; ++sweep) {
        if (sweep == 1) {
            delta += __shfl_xor(delta, 32, 64);
            __builtin_amdgcn_s_barrier();  // every wave is done reading the last tile of sweep 1
        }
        issue_tile<HDP>(Kb, a.ldkv, 0, a.Nk, a.hd, smem, wid, lane);
        issue_tile<HDP>(Vb, a.ldkv, 0, a.Nk, a.hd, smem + TILE, wid, lane);
        for (int kt = 0; kt < nkt; ++kt) {
            wait_dma();
            __builtin_amdgcn_s_barrier();
            const char* Ks = smem + (kt & 1) * 2 * TILE;
            const char* Vs = Ks + TILE;
            if (kt + 1 < nkt) {
                char* nx = smem + ((kt + 1) & 1) * 2 * TILE;
                issue_tile<HDP>(Kb, a.ldkv, (kt + 1) * TR, a.Nk, a.hd, nx, wid, lane);
                issue_tile<HDP>(Vb, a.ldkv, (kt + 1) * TR, a.Nk, a.hd, nx + TILE, wid, lane);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s = product_T<HDP>(Ks, 32 * kb, qf, lane);    // S^T
                const f32x16 dp = product_T<HDP>(Vs, 32 * kb, gf, lane);  // dP^T = V dO^T
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool dead = kt == nkt - 1 && kt * TR + 32 * kb + frag_row(r, half) >= a.Nk;
                    const float p32 = dead ? 0.f : exp2_raw(logit2<MODE>(s[r], sc) - m) * linv;
                    const float d16 = rnd16(dp[r]);
                    if (sweep == 0) delta = fmaf(d16, p32, delta);
                    else s[r] = dlogit<MODE>(p32 * (d16 - delta), sc);
                }
                if (sweep == 1) accumulate_T<HDP>(dq, Ks, 32 * kb, s, lane);   // dq^T += K^T dS^T
            }
        }
    }
    if (q < a.Nq) {
        store_T<HDP>(dq, a.dq + ((long)b * a.Nq + q) * a.lddq + h * a.hd, a.hd, half);
        if (half == 0) a.delta[((long)b * a.H + h) * a.Nq + q] = delta;
    }
}

// ---- backward, keys resident: dk = rnd(dS^T q), dv = rnd(P^T dO); queries streamed --------------------------------------------
template <int HDP, int MODE>
__global__ __launch_bounds__(NW * 64) void attn16_train_bwd_dkv_kernel(Ptrs a) {
    using G = Geo<HDP>;
    constexpr int TILE = G::TILE;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE];   // [stage][Q | dO]
    __shared__ __attribute__((aligned(16))) float stats[2][3][TR];                               // [stage][m | linv | delta][query of the tile]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, key = (blockIdx.x * NW + wid) * 32 + l31;
    const __bf16* Qb = a.q + b * a.q_bs + h * a.hd;
    const __bf16* Gb = a.d_o + (long)b * a.Nq * a.lddo + h * a.hd;
    const float* Mb = a.m + ((long)b * a.H + h) * a.Nq;
    const float* Lb = a.linv + ((long)b * a.H + h) * a.Nq;
    const float* Db = a.delta + ((long)b * a.H + h) * a.Nq;
    bf16x8 kf[G::KC], vf[G::KC];
    zero_lds<4 * TILE>(smem, tid);
    load_resident<HDP>(kf, a.k + (long)b * a.Nk * a.ldkv + h * a.hd, a.ldkv, key, a.Nk, a.hd, half);
    load_resident<HDP>(vf, a.v + (long)b * a.Nk * a.ldkv + h * a.hd, a.ldkv, key, a.Nk, a.hd, half);
    const bool dead_key = key >= a.Nk;
    const int nqt = (a.Nq + TR - 1) / TR;
    const Scales sc{a.c1, a.c2, a.c1 * a.c2, a.c1 * a.c2 * LOG2E};
    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int c = 0; c < G::DB; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[c][r] = dv[c][r] = 0.f;
    auto stage_stats = [&](int qt, int st) {
        if (tid < TR) {
            const int qi = min(qt * TR + tid, a.Nq - 1);
            stats[st][0][tid] = Mb[qi];
            stats[st][1][tid] = Lb[qi];
            stats[st][2][tid] = Db[qi];
        }
    };
    issue_tile<HDP>(Qb, a.ldq, 0, a.Nq, a.hd, smem, wid, lane);
    issue_tile<HDP>(Gb, a.lddo, 0, a.Nq, a.hd, smem + TILE, wid, lane);
    stage_stats(0, 0);
    for (int qt = 0; qt < nqt; ++qt) {
        wait_dma();
        __syncthreads();   // (also publishes the ds_write of the statistics)
        const char* Qs = smem + (qt & 1) * 2 * TILE;
        const char* Gs = Qs + TILE;
        if (qt + 1 < nqt) {
            char* nx = smem + ((qt + 1) & 1) * 2 * TILE;
            issue_tile<HDP>(Qb, a.ldq, (qt + 1) * TR, a.Nq, a.hd, nx, wid, lane);
            issue_tile<HDP>(Gb, a.lddo, (qt + 1) * TR, a.Nq, a.hd, nx + TILE, wid, lane);
            stage_stats(qt + 1, (qt + 1) & 1);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 s = product_T<HDP>(Qs, 32 * qb, kf, lane);          // S: rows = queries (registers), lane = key
            f32x16 dp = product_T<HDP>(Gs, 32 * qb, vf, lane);         // dP
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qo = 32 * qb + 8 * g + 4 * half;        // the 4 consecutive queries of registers 4 g .. 4 g + 3
                const float4 m4 = *reinterpret_cast<const float4*>(&stats[qt & 1][0][qo]);
                const float4 l4 = *reinterpret_cast<const float4*>(&stats[qt & 1][1][qo]);
                const float4 d4 = *reinterpret_cast<const float4*>(&stats[qt & 1][2][qo]);
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * g + u;
                    // (a lane past the last key works on a repeated row and is never stored: only ragged QUERY rows need zeros)
                    const bool dead = qt == nqt - 1 && qt * TR + qo + u >= a.Nq;
                    const float p32 = dead ? 0.f : exp2_raw(logit2<MODE>(s[r], sc) - mm[u]) * ll[u];
                    const float d16 = rnd16(dp[r]);
                    dp[r] = dlogit<MODE>(p32 * (d16 - dd[u]), sc);   // dS (0 where p32 is 0)
                    s[r] = rnd16(p32);                               // P
                }
            }
            accumulate_T<HDP>(dv, Gs, 32 * qb, s, lane);    // dV^T += dO^T P
            accumulate_T<HDP>(dk, Qs, 32 * qb, dp, lane);   // dK^T += Q^T dS
        }
    }
    if (!dead_key) {
        store_T<HDP>(dk, a.dk + ((long)b * a.Nk + key) * a.lddkv + h * a.hd, a.hd, half);
        store_T<HDP>(dv, a.dv + ((long)b * a.Nk + key) * a.lddkv + h * a.hd, a.hd, half);
    }
}

int check(const dkd_attn16_args* g, bool bwd) {
    if (!g || !g->q || !g->k || !g->v || !g->m || !g->linv || g->B <= 0 || g->H <= 0 || g->Nq <= 0 || g->Nk <= 0) return DKD_ERR_BAD_ARG;
    if ((g->hd & 7) || g->hd < 16 || g->hd > 96) return DKD_ERR_UNSUPPORTED;  // 8-wide chunks of a 32-, 64- or 96-wide tile (MiT: 40, 80)
    if (g->H > 65535 || g->B > 65535) return DKD_ERR_UNSUPPORTED;
    const long w = (long)g->H * g->hd;
    if ((g->ldq & 7) || (g->ldkv & 7) || g->ldq < w || g->ldkv < w) return DKD_ERR_ALIGN;
    if (!dkd_aligned16(g->q) || !dkd_aligned16(g->k) || !dkd_aligned16(g->v)) return DKD_ERR_ALIGN;
    if (!bwd) {
        if (!g->o || (g->ldo & 3) || g->ldo < w || (((uintptr_t)g->o) & 7)) return DKD_ERR_ALIGN;
    } else {
        if (!g->d_o || !g->dq || !g->dk || !g->dv || !g->delta) return DKD_ERR_BAD_ARG;
        if ((g->lddo & 7) || (g->lddq & 3) || (g->lddkv & 3) || g->lddo < w || g->lddq < w || g->lddkv < w) return DKD_ERR_ALIGN;
        if (!dkd_aligned16(g->d_o) || (((uintptr_t)g->dq) & 7) || (((uintptr_t)g->dk) & 7) || (((uintptr_t)g->dv) & 7)) return DKD_ERR_ALIGN;
    }
    return DKD_OK;
}
Ptrs pack(const dkd_attn16_args* g) {
    Ptrs a;
    a.q = reinterpret_cast<const __bf16*>(g->q), a.k = reinterpret_cast<const __bf16*>(g->k), a.v = reinterpret_cast<const __bf16*>(g->v);
    a.d_o = reinterpret_cast<const __bf16*>(g->d_o);
    a.o = reinterpret_cast<__bf16*>(g->o), a.dq = reinterpret_cast<__bf16*>(g->dq), a.dk = reinterpret_cast<__bf16*>(g->dk);
    a.dv = reinterpret_cast<__bf16*>(g->dv);
    a.m = g->m, a.linv = g->linv, a.delta = g->delta;
    a.ldq = g->ldq, a.ldkv = g->ldkv, a.ldo = g->ldo, a.lddo = g->lddo, a.lddq = g->lddq, a.lddkv = g->lddkv;
    a.q_bs = g->q_shared ? 0 : (long)g->Nq * g->ldq;
    a.B = g->B, a.H = g->H, a.Nq = g->Nq, a.Nk = g->Nk, a.hd = g->hd, a.c1 = g->c1, a.c2 = g->c2;
    return a;
}

}  // namespace

// MODE of the logit chain (see logit2): 2 = c1 a power of two and c2 == 1, 1 = c1 a power of two, 0 = general
static int scale_mode(float c1, float c2) {
    int e;
    const bool pow2 = c1 > 0.f && frexpf(c1, &e) == 0.5f;
    return pow2 ? (c2 == 1.0f ? 2 : 1) : 0;
}
#define DKD_A16_MODE(mode, K, grid, a)                                                                                      \
    do {                                                                                                                    \
        if (a.hd > 64) {   /* (no power-of-two scale at these widths: the general logit chain) */                          \
            hipLaunchKernelGGL((K<96, 0>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);                                 \
        } else if (a.hd > 32) {                                                                                             \
            if ((mode) == 2) hipLaunchKernelGGL((K<64, 2>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);                \
            else if ((mode) == 1) hipLaunchKernelGGL((K<64, 1>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);           \
            else hipLaunchKernelGGL((K<64, 0>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);                            \
        } else {                                                                                                            \
            if ((mode) == 2) hipLaunchKernelGGL((K<32, 2>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);                \
            else if ((mode) == 1) hipLaunchKernelGGL((K<32, 1>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);           \
            else hipLaunchKernelGGL((K<32, 0>), grid, dim3(NW * 64), 0, (hipStream_t)stream, a);                            \
        }                                                                                                                   \
    } while (0)

extern "C" int dkd_attn16_train_fwd(const dkd_attn16_args* g, void* stream) {
    const int rc = check(g, false);
    if (rc) return rc;
    const Ptrs a = pack(g);
    const int mode = scale_mode(g->c1, g->c2);
    DKD_A16_MODE(mode, attn16_train_fwd_kernel, dim3((g->Nq + 32 * NW - 1) / (32 * NW), g->H, g->B), a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_attn16_train_bwd(const dkd_attn16_args* g, void* stream) {
    const int rc = check(g, true);
    if (rc) return rc;
    const Ptrs a = pack(g);
    const int mode = scale_mode(g->c1, g->c2);
    DKD_A16_MODE(mode, attn16_train_bwd_dq_kernel, dim3((g->Nq + 32 * NW - 1) / (32 * NW), g->H, g->B), a);
    DKD_LAUNCH_CHECK();
    DKD_A16_MODE(mode, attn16_train_bwd_dkv_kernel, dim3((g->Nk + 32 * NW - 1) / (32 * NW), g->H, g->B), a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
