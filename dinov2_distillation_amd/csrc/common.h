// Shared device/host helpers for the gfx950 (CDNA4, MI355X) kernels of the distillation-step hot path.
// Wavefront = 64 lanes everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DKD_WAVE 64

// C-ABI status codes (0 == hipSuccess). Negative values are argument errors detected on the host.
#define DKD_OK 0
#define DKD_ERR_BAD_ARG (-1)
#define DKD_ERR_ALIGN (-2)
#define DKD_ERR_UNSUPPORTED (-3)

#define DKD_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int dkd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool dkd_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact (erf) GELU and its derivative, as nn.GELU() / F.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
