#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters, float a0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(a0 * 0.5f + threadIdx.x * 2e-3f - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        const int iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_f32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = 2.0 * 32 * 32 * 2 * 4.0 * iters * blocks * 4;
            if (rep) printf("f32 mfma 32x32x2: %d blocks x 4 waves, 4 acc: %.1f TFLOP/s (%.2f ms)\n", blocks, fl / ms / 1e9, ms);
            hipEventRecord(e0); hipLaunchKernelGGL(k_bf16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            fl = 2.0 * 32 * 32 * 16 * 4.0 * iters * blocks * 4;
            if (rep) printf("bf16 mfma 32x32x16: %d blocks x 4 waves, 4 acc: %.1f TFLOP/s (%.2f ms)\n", blocks, fl / ms / 1e9, ms);
        }
    }
    return 0;
}
