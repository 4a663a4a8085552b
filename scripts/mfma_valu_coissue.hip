// How many plain VALU instructions does one v_mfma_f32_32x32x16_bf16 cover on gfx950?  (design input of csrc/gemm_x3.hip)
// One workgroup per CU, W waves per SIMD; each wave runs a loop of {1 MFMA + N VALU ops of kind KIND}, all independent of the MFMA;
// prints shader cycles per MFMA for N = 0 .. 12.   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_coissue.hip -o /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_sub_f32   1: v_and_b32   2: v_cvt_pk_bf16_f32   3: v_perm_b32   4: v_lshlrev_b32   5: the real split mix (cvt, shl, and, sub, sub)
template <int N, int KIND, bool DEP>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e + 1); }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float x[12];
    unsigned u[12];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 d[6];
    for (int i = 0; i < 6; ++i) d[i] = f2{threadIdx.x * 0.11f + i, threadIdx.x * 0.13f - i};
    for (int i = 0; i < 12; ++i) { x[i] = threadIdx.x * 0.37f + i; u[i] = threadIdx.x * 77u + i; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[DEP ? 0 : q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[DEP ? 0 : q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (KIND == 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % 12]));
                if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
                if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) % 12]));
                if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(u[(i + 2) % 12]), "s"(0x07060302u));
                if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i % 6]) : "v"(d[(i + 1) % 6]));
                if (KIND == 7) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) % 12]));
                if (KIND == 8) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) % 12]));
                if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i % 6]) : "v"(d[(i + 1) % 6]));
                if (KIND == 10) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(u[(i + 2) % 12]));
                if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 1) % 12]));
                if (KIND == 5) {
                    switch (i % 5) {
                        case 0: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) % 12])); break;
                        case 1: asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i])); break;
                        case 2: asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i])); break;
                        default: asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) % 12])); break;
                    }
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 12; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 6; ++i) s += d[i][0] + d[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int N, int KIND, bool DEP>
double run(int waves_per_simd, float* out, long long* cyc) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<N, KIND, DEP>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<N, KIND, DEP>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    // cycles per MFMA *per SIMD*: a wave's wall cycles / its MFMAs / waves sharing the SIMD
    return s / h.size() / (iters * 4.0) / waves_per_simd;
}

template <int KIND, bool DEP>
void sweep(const char* name, float* out, long long* cyc) {
    for (int w = 1; w <= 2; ++w) {
        printf("%-28s %s  %d wave/SIMD  cycles per MFMA per SIMD at N = 0,2,4,6,7,8,10,12 VALU: ", name, DEP ? "dependent acc " : "4 accumulators", w);
        printf("%.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n", run<0, KIND, DEP>(w, out, cyc), run<2, KIND, DEP>(w, out, cyc), run<4, KIND, DEP>(w, out, cyc),
               run<6, KIND, DEP>(w, out, cyc), run<7, KIND, DEP>(w, out, cyc), run<8, KIND, DEP>(w, out, cyc), run<10, KIND, DEP>(w, out, cyc),
               run<12, KIND, DEP>(w, out, cyc));
    }
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    sweep<0, false>("v_sub_f32", out, cyc);
    sweep<1, false>("v_and_b32", out, cyc);
    sweep<2, false>("v_cvt_pk_bf16_f32", out, cyc);
    sweep<3, false>("v_perm_b32", out, cyc);
    sweep<4, false>("v_lshlrev_b32", out, cyc);
    sweep<5, false>("split mix", out, cyc);
    sweep<6, false>("v_pk_add_f32", out, cyc);
    sweep<7, false>("v_dot2c_f32_bf16", out, cyc);
    sweep<8, false>("v_dot2_f32_bf16", out, cyc);
    sweep<9, false>("v_pk_mul_f32", out, cyc);
    sweep<10, false>("v_bfi_b32", out, cyc);
    sweep<11, false>("v_fma_f32", out, cyc);
    return 0;
}
