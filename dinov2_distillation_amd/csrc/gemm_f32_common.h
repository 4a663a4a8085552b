// Shared by the fp32-operand GEMM kernels (gemm_f32.hip: exact fp32 MFMA; gemm_x3.hip: bf16x3 split on the bf16 matrix cores):
// the fused epilogue of dkd_gemm_f32 (include/dkd_hip.h), the XCD tile order and the split-K fold launcher.
#pragma once
#include "common.h"
#include "dkd_hip.h"

struct DkdF32Epilogue {
    const float* bias;
    const float* colscale;
    const float* residual;
    float* preact;
    long ldr;
    int res_mod;
    int act;
    int row_grp, row_grp_stride, row_off;
    float alpha;
    int order;  // tile order experiment (DKD_TILE_ORDER, scripts/pmc_tile_order.sh): 0 = xcd_remap (shipped)
};
typedef DkdF32Epilogue Epilogue;

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DKD_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DKD_ACT_GELU) return gelu_f(v);
    return v;
}

__device__ __forceinline__ void epilogue_store(const Epilogue& e, float* C, long ldc, int m, int n,
                                               float acc) {
    float v = acc * e.alpha;
    if (e.bias) v += e.bias[n];
    long orow = m;
    if (e.row_grp > 0) orow = (long)(m / e.row_grp) * e.row_grp_stride + (m % e.row_grp) + e.row_off;
    if (e.preact) e.preact[orow * ldc + n] = v;
    v = apply_act(v, e.act);
    if (e.colscale) v *= e.colscale[n];
    if (e.residual) {
        const long rr = e.res_mod > 0 ? (long)(m % e.res_mod) : orow;
        v += e.residual[rr * e.ldr + n];
    }
    C[orow * ldc + n] = v;
}

// XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous run of
// logical tiles (bn fastest) -> the tiles that share an A row-panel hit the same L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, j = bid / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

__device__ __forceinline__ void epilogue_store4(const Epilogue& e, float* C, long ldc, int m, int n, int N, f32x4 acc, bool vec) {
    if (!vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) epilogue_store(e, C, ldc, m, n + r, acc[r]);
        return;
    }
    float4 v = make_float4(acc[0] * e.alpha, acc[1] * e.alpha, acc[2] * e.alpha, acc[3] * e.alpha);
    if (e.bias) {
        const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    long orow = m;
    if (e.row_grp > 0) orow = (long)(m / e.row_grp) * e.row_grp_stride + (m % e.row_grp) + e.row_off;
    if (e.preact) *reinterpret_cast<float4*>(e.preact + orow * ldc + n) = v;
    v.x = apply_act(v.x, e.act); v.y = apply_act(v.y, e.act); v.z = apply_act(v.z, e.act); v.w = apply_act(v.w, e.act);
    if (e.colscale) {
        const float4 c = *reinterpret_cast<const float4*>(e.colscale + n);
        v.x *= c.x; v.y *= c.y; v.z *= c.z; v.w *= c.w;
    }
    if (e.residual) {
        const long rr = e.res_mod > 0 ? (long)(m % e.res_mod) : orow;
        const float4 r = *reinterpret_cast<const float4*>(e.residual + rr * e.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(C + orow * ldc + n) = v;
}

}  // namespace

// gemm_f32.hip: fold `splits` K-slab partial sums [splits][M][N] (+ [splits][M] row sums behind them) of g->workspace into g->C
// through the epilogue, in a fixed order
int dkd_f32_splitk_reduce(const dkd_gemm_args* g, const Epilogue& epi, int splits, hipStream_t st);
// gemm_x3.hip: the bf16x3 split-precision kernels (LDS-DMA ring); DKD_ERR_UNSUPPORTED = take another path
int dkd_x3_launch(const dkd_gemm_args* g, const Epilogue& epi, hipStream_t st);
long dkd_x3_workspace_bytes(int M, int N, int K, int a_layout, int b_layout);
bool dkd_x3_conv_ok(int B, int H, int W, int C, int O, int k, int Ho, int Wo);
int dkd_x3_conv_forward(const float* x, const float* w, float* out, const float* bias, float* workspace, int max_splits, int B, int H, int W,
                        int C, int O, int k, int stride, int pad, int Ho, int Wo, hipStream_t st);
