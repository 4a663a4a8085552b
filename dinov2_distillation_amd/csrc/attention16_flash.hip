// Flash-style attention forward on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) for the frozen DINOv2 teacher under the
// reference's default precision=16 policy (train.py:263 == torch autocast): q, k, v are the bf16 column slices of the fused qkv
// Linear's 16-bit output, the result is the 16-bit tensor the proj Linear consumes.  head_dim 64 (every DINOv2 model), forward
// only (the teacher's own forward needs no gradient; the re-use of its blocks on student tokens goes through ops.attention16).
//
// Rounding points (what eager attention under autocast rounds, as csrc/attention.hip's round_bf16 mode of the fp32 kernel):
//   S = rnd(q k^T)        a 16-bit matmul result; * head_dim^-0.5 is exact (a power of two for 64-wide heads)
//   softmax in fp32, online over 64-key tiles; p = rnd(exp(S - running max)) is the 16-bit operand of the second matmul
//   O = (sum p v) / sum p  fp32 accumulation of exact bf16 x bf16 products, one rounding at the store
//
// Formulation ("everything transposed", attention.hip): S^T = K Q^T puts the query on the lane - row max / sum are 32 in-lane values
// plus one lane^32 exchange - and the S^T accumulator registers, packed to bf16 eight at a time, ARE the B operand of
// O^T += V^T P^T: register r of lane (half, q) is key (r & 3) + 8 (r >> 2) + 4 half of its 32-key block, i.e. the eight registers
// 8 b .. 8 b + 7 are keys 16 b + 4 half + {0..3} and 16 b + 8 + 4 half + {0..3} - exactly two transposing LDS reads
// (ds_read_b64_tr_b16) of four consecutive V rows each, so no key permutation of the tiles is needed.
// K / V tiles (64 keys x 64 d, 8 KiB each) go global -> LDS by LDS-DMA into two stages (one barrier per tile); the bank swizzle
// sits on the source address (lane-linear LDS image): K rows XOR their 16-byte chunk with (row / 2) % 8 (conflict-free
// ds_read_b128 fragments over 16 consecutive rows), V rows with 2 ((row / 2) % 4).
#include <cstdlib>

#include "common.h"
#include "dkd_hip.h"
#include "gemm_bf16_common.h"

namespace {
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr float LOG2E = 1.4426950408889634f;
constexpr int HD = 64, KT = 64, ROWB = HD * 2, TILE = KT * ROWB, STAGE = 2 * TILE;

__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int v_swz(int row) { return ((row >> 1) & 3) << 1; }

struct FlashArgs {
    const __bf16 *q, *k, *v;
    __bf16* o;
    long ld, ldo;   // row pitch (elements) of the q / k / v slices and of o
    int B, H, N;
    float scale;
};

// one wave's 8 / NW pieces (1 KiB each: 8 rows x 128 B) of a 64-row tile: piece = wid + NW p
template <int NW>
__device__ __forceinline__ void issue_tile(const __bf16* __restrict__ base, long ld, int key0, int N, char* __restrict__ S, int wid,
                                           int lane, bool is_v) {
#pragma unroll
    for (int p = 0; p < 8 / NW; ++p) {
        const int piece = wid + NW * p, row = 8 * piece + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ (is_v ? v_swz(row) : k_swz(row));
        const int key = min(key0 + row, N - 1);   // rows past the last key repeat it: their logits are masked, their p is 0
        const __bf16* src = base + (long)key * ld + 8 * chunk;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(S + piece * 1024), 16, 0, 0);
    }
}

// NW waves of 32 queries each share the K / V tiles
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_flash_fwd_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];  // [stage][K | V]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * NW + wid) * 32, q = q0 + l31;
    const long boff = (long)b * a.N * a.ld + h * HD;
    const __bf16* Kb = a.k + boff;
    const __bf16* Vb = a.v + boff;

    // B operand of S^T = K Q^T: this lane's query row, d in [16 kc + 8 half, +8)
    bf16x8 qf[4];
    {
        const __bf16* qr = a.q + boff + (long)min(q, a.N - 1) * a.ld + 8 * half;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) qf[kc] = *reinterpret_cast<const bf16x8*>(qr + 16 * kc);
    }
    f32x16 o[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sc2 = a.scale * LOG2E;
    const int nkt = (a.N + KT - 1) / KT;

    issue_tile<NW>(Kb, a.ld, 0, a.N, smem, wid, lane, false);
    issue_tile<NW>(Vb, a.ld, 0, a.N, smem + TILE, wid, lane, true);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile kt is visible to every wave; nobody reads the other stage any more
        const char* Ks = smem + (kt & 1) * STAGE;
        const char* Vs = Ks + TILE;
        if (kt + 1 < nkt) {
            char* nx = smem + ((kt + 1) & 1) * STAGE;
            issue_tile<NW>(Kb, a.ld, (kt + 1) * KT, a.N, nx, wid, lane, false);
            issue_tile<NW>(Vb, a.ld, (kt + 1) * KT, a.N, nx + TILE, wid, lane, true);
        }
        // ---- S^T (two 32-key blocks) ----
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const int row = 32 * kb + l31;
            const char* kr = Ks + row * ROWB;
            const int sw = k_swz(row);
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kr + 16 * ((2 * kc + half) ^ sw));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kc], s[kb], 0, 0, 0);
            }
        }
        // ---- online softmax over the 64 keys of the tile (the query sits on the lane; the other half holds the other 32 keys) ----
        const bool last = kt == nkt - 1;
        float mt = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = rnd16(s[kb][r]) * sc2;   // the 16-bit logit, in log2 units
                if (last && kt * KT + 32 * kb + frag_row(r, half) >= a.N) v = -INFINITY;
                s[kb][r] = v;
                mt = fmaxf(mt, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);
        const float alpha = exp2_raw(m - mn);
        m = mn;
        float ls = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = rnd16(exp2_raw(s[kb][r] - mn));
                s[kb][r] = p;
                ls += p;
            }
        l = l * alpha + ls;   // (each half keeps the sum over its own keys; the halves are added at the end)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] *= alpha;
        // ---- O^T += V^T P^T ----
        const int cb = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int b16 = 0; b16 < 2; ++b16) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[kb][8 * b16 + e];   // (exact: the values are 16-bit already)
                const int r_lo = 32 * kb + 16 * b16 + 4 * half + qq, r_hi = r_lo + 8;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ch = 4 * c + 2 * cb + (pp >> 1);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s16x4*)(Vs + r_lo * ROWB + 16 * (ch ^ v_swz(r_lo)) + 8 * (pp & 1)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (lds_s16x4*)(Vs + r_hi * ROWB + 16 * (ch ^ v_swz(r_hi)) + 8 * (pp & 1)));
                    union { s16x4 v[2]; bf16x8 f; } u;
                    u.v[0] = lo;
                    u.v[1] = hi;
                    o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.f, pf, o[c], 0, 0, 0);
                }
            }
    }
    l += __shfl_xor(l, 32, 64);
    if (q < a.N) {
        const float inv = 1.0f / l;
        __bf16* orow = a.o + ((long)b * a.N + q) * a.ldo + h * HD;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * c + 8 * g + 4 * half;
                const bf16x4 v = {(__bf16)(o[c][4 * g] * inv), (__bf16)(o[c][4 * g + 1] * inv), (__bf16)(o[c][4 * g + 2] * inv),
                                  (__bf16)(o[c][4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4*>(orow + d) = v;
            }
    }
}

}  // namespace

extern "C" int dkd_attn16_flash_fwd(const void* q, const void* k, const void* v, void* o, long ld, long ldo, int B, int H, int N, int hd,
                                    float scale, void* stream) {
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || N <= 0) return DKD_ERR_BAD_ARG;
    if (hd != HD) return DKD_ERR_UNSUPPORTED;
    if ((ld & 7) || (ldo & 3) || ld < (long)H * HD || ldo < (long)H * HD) return DKD_ERR_ALIGN;
    if (!dkd_aligned16(q) || !dkd_aligned16(k) || !dkd_aligned16(v) || (((uintptr_t)o) & 7)) return DKD_ERR_ALIGN;
    if (H > 65535 || B > 65535) return DKD_ERR_UNSUPPORTED;
    FlashArgs a{reinterpret_cast<const __bf16*>(q), reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v),
                reinterpret_cast<__bf16*>(o), ld, ldo, B, H, N, scale};
    // 4 waves (128 queries) per workgroup share each K / V tile.  A 2-wave form (less query padding for the teacher's 257 tokens:
    // 5 x 64 instead of 3 x 128) measured SLOWER on every shape (44.8 vs 39.8 us at bs 64 x 6 heads x 257; 369 vs 280 us at 1370
    // tokens): kept behind DKD_FLASH16_WAVES=2 for A/B runs only.
    static const int forced = getenv("DKD_FLASH16_WAVES") ? atoi(getenv("DKD_FLASH16_WAVES")) : 0;
    if (forced == 2)
        hipLaunchKernelGGL(attn16_flash_fwd_kernel<2>, dim3((N + 63) / 64, H, B), dim3(128), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(attn16_flash_fwd_kernel<4>, dim3((N + 127) / 128, H, B), dim3(256), 0, (hipStream_t)stream, a);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
