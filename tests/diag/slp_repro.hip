// Standalone reproducer for the `-fno-slp-vectorize` finding (NOTES 9.16): no library, one file.
//
//   hipcc -O3 --offload-arch=gfx950 tests/diag/slp_repro.hip -o /tmp/slp_repro && /tmp/slp_repro
//
// Round 2 saw resize_fwd_kernel built WITH hipcc's SLP vectoriser return different bits for the same input while
// v_mfma_f32_32x32x16_bf16 waves of another stream were co-resident; the same source without packed math was immune.  The ISA diff
// (tests/diag/slp_isa) showed the same arithmetic order in both builds, the packed one made of
//     v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,1]   (lo = hi = src0.lo * src1.hi: the broadcast bilinear weights)
//     v_pk_mul_f32 / v_pk_fma_f32                      (the four-tap blend of two channels at a time)
// so either packed-fp32 op_sel reads misbehave beside bf16 MFMA traffic on the same SIMD (hardware / code generation), or the library
// had a timing-dependent race that the changed code hid.  This program isolates the first possibility: a kernel made of exactly those
// packed forms (inline asm, so no compiler flag can change it) against its scalar twin (v_mul_f32 / v_fma_f32 in the same order), both
// run many times beside a bf16-MFMA co-runner on a second stream with all CUs shared, every output word compared with the output of
// the same kernel run alone.  Prints the number of differing words per arm; exit code 0 either way (it is a measurement).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } \
    } while (0)

// the bilinear blend of two channels at a time, as the SLP build of resize_fwd_kernel does it: fx, fy in [0, 1)
//   w = (fx, fy); iw = 1 - w; w11 = w.x*w.y, w01 = w.x*iw.y (per the op_sel products), w00 = iw.x*iw.y, w10 = iw.x*w.y
//   out = a*w00 + b*w01 + c*w10 + d*w11, the products formed first for b (mul), then fma a, c, d - the order of the shipped ISA
__global__ void packed_kernel(const f2* __restrict__ a, const f2* __restrict__ b, const f2* __restrict__ c, const f2* __restrict__ d,
                              const f2* __restrict__ w, f2* __restrict__ out, int n, int reps) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f2 va = a[i], vb = b[i], vc = c[i], vd = d[i], vw = w[i], iw, w11, w01, w00, w10, acc;
        for (int r = 0; r < reps; ++r) {
            asm volatile(
                "v_pk_add_f32 %0, %6, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"   // iw = 1 - w
                "s_nop 0\n\t"
                "v_pk_mul_f32 %1, %6, %6 op_sel:[0,1] op_sel_hi:[0,1]\n\t"                 // w11 = w.lo * w.hi (both halves)
                "v_pk_mul_f32 %2, %6, %0 op_sel:[0,1] op_sel_hi:[0,1]\n\t"                 // w01 = w.lo * iw.hi
                "v_pk_mul_f32 %3, %0, %0 op_sel:[0,1] op_sel_hi:[0,1]\n\t"                 // w00 = iw.lo * iw.hi
                "v_pk_mul_f32 %4, %0, %6 op_sel:[0,1] op_sel_hi:[0,1]\n\t"                 // w10 = iw.lo * w.hi
                "s_nop 0\n\t"
                "v_pk_mul_f32 %5, %8, %2\n\t"                                               // acc = b * w01
                "s_nop 0\n\t"
                "v_pk_fma_f32 %5, %7, %3, %5\n\t"                                           // + a * w00
                "s_nop 0\n\t"
                "v_pk_fma_f32 %5, %9, %4, %5\n\t"                                           // + c * w10
                "s_nop 0\n\t"
                "v_pk_fma_f32 %5, %1, %10, %5\n\t"                                          // + w11 * d
                : "=&v"(iw), "=&v"(w11), "=&v"(w01), "=&v"(w00), "=&v"(w10), "=&v"(acc)
                : "v"(vw), "v"(va), "v"(vb), "v"(vc), "v"(vd));
            va = acc;  // chain the repetitions (every one is data dependent on the previous)
        }
        out[i] = acc;
    }
}

__global__ void scalar_kernel(const f2* __restrict__ a, const f2* __restrict__ b, const f2* __restrict__ c, const f2* __restrict__ d,
                              const f2* __restrict__ w, f2* __restrict__ out, int n, int reps) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f2 va = a[i], acc;
        const f2 vb = b[i], vc = c[i], vd = d[i], vw = w[i];
        for (int r = 0; r < reps; ++r) {
            const float ix = 1.0f - vw[0], iy = 1.0f - vw[1];
            float w11, w01, w00, w10, lo, hi;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w11) : "v"(vw[0]), "v"(vw[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w01) : "v"(vw[0]), "v"(iy));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w00) : "v"(ix), "v"(iy));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w10) : "v"(ix), "v"(vw[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(vb[0]), "v"(w01));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(vb[1]), "v"(w01));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(va[0]), "v"(w00));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(va[1]), "v"(w00));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(vc[0]), "v"(w10));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(vc[1]), "v"(w10));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(w11), "v"(vd[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(w11), "v"(vd[1]));
            acc = f2{lo, hi};
            va = acc;
        }
        out[i] = acc;
    }
}

// Victim 3: the bilinear resize itself (the arithmetic of csrc/conv.hip resize_fwd_kernel), compiled by hipcc -O3 WITH its SLP
// vectoriser (this file is built without -fno-slp-vectorize): the packed forms are whatever the compiler makes of it.
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& lam) {
    float s = ((float)o + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}
__global__ void resize_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int cq = C >> 2;
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    const long total = (long)B * Ho * Wo * cq;
    const float4* in4 = reinterpret_cast<const float4*>(in);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int c4 = (int)(r % cq); r /= cq;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, sy, Hi, y0, y1, ly);
        src_index(ox, sx, Wi, x0, x1, lx);
        const long base = (long)b * Hi * Wi;
        const float4 a = in4[(base + (long)y0 * Wi + x0) * cq + c4], bq = in4[(base + (long)y0 * Wi + x1) * cq + c4];
        const float4 c = in4[(base + (long)y1 * Wi + x0) * cq + c4], d = in4[(base + (long)y1 * Wi + x1) * cq + c4];
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        float4 o;
        o.x = w00 * a.x + w01 * bq.x + w10 * c.x + w11 * d.x;
        o.y = w00 * a.y + w01 * bq.y + w10 * c.y + w11 * d.y;
        o.z = w00 * a.z + w01 * bq.z + w10 * c.z + w11 * d.z;
        o.w = w00 * a.w + w01 * bq.w + w10 * c.w + w11 * d.w;
        reinterpret_cast<float4*>(out)[idx] = o;
    }
}

// Co-runner 2: what a bf16 GEMM does besides MFMAs - operand fragments streamed from global memory through LDS (ds_write_b128 /
// ds_read_b128, a barrier per tile), two bf16 MFMAs per fragment pair
__global__ __launch_bounds__(256) void gemm_like_corunner(const bf16x8* __restrict__ src, float* sink, int tiles, int nsrc) {
    __shared__ bf16x8 lds[2][256];
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    for (int t = 0; t < tiles; ++t) {
        const bf16x8 g0 = src[p % nsrc], g1 = src[(p + 7919) % nsrc];
        p += 104729;
        lds[0][threadIdx.x] = g0;
        lds[1][threadIdx.x] = g1;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x8 a = lds[0][(threadIdx.x + 64 * k) & 255], b = lds[1][(threadIdx.x + 64 * k + 32) & 255];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Co-runner 1: bf16 MFMA 32x32x16 back to back on every SIMD from registers only, for `iters` MFMAs per wave
__global__ __launch_bounds__(256) void mfma_corunner(float* sink, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (e + 1)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;  // (keeps the loop; never true)
}

int main() {
    const int n = 1 << 22, reps = 8, rounds = 40;
    std::vector<f2> h(n);
    f2 *a, *b, *c, *d, *w, *out, *ref;
    float* sink;
    CK(hipMalloc(&a, n * sizeof(f2))); CK(hipMalloc(&b, n * sizeof(f2))); CK(hipMalloc(&c, n * sizeof(f2))); CK(hipMalloc(&d, n * sizeof(f2)));
    CK(hipMalloc(&w, n * sizeof(f2))); CK(hipMalloc(&out, n * sizeof(f2))); CK(hipMalloc(&ref, n * sizeof(f2)));
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    srand(1234);
    for (f2* p : {a, b, c, d}) {
        for (int i = 0; i < n; ++i) h[i] = f2{(float)rand() / RAND_MAX * 2.f - 1.f, (float)rand() / RAND_MAX * 2.f - 1.f};
        CK(hipMemcpy(p, h.data(), n * sizeof(f2), hipMemcpyHostToDevice));
    }
    for (int i = 0; i < n; ++i) h[i] = f2{(float)rand() / RAND_MAX, (float)rand() / RAND_MAX};
    CK(hipMemcpy(w, h.data(), n * sizeof(f2), hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    // operands of the GEMM-like co-runner (finite bf16 values) and of the compiled resize victim
    const int nsrc = 1 << 20;
    bf16x8* gsrc;
    CK(hipMalloc(&gsrc, nsrc * sizeof(bf16x8)));
    {
        std::vector<unsigned short> hb((size_t)nsrc * 8);
        for (auto& v : hb) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));  // bf16 in [0.0078, 0.03)
        CK(hipMemcpy(gsrc, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    }
    const int RB = 32, RH = 7, RW = 7, RC = 1024, RO = 16;   // the round-2 case: 32 x 7 x 7 x 1024 -> 16 x 16
    const size_t rin_n = (size_t)RB * RH * RW * RC, rout_n = (size_t)RB * RO * RO * RC;
    float *rin, *rout, *rref;
    CK(hipMalloc(&rin, rin_n * 4)); CK(hipMalloc(&rout, rout_n * 4)); CK(hipMalloc(&rref, rout_n * 4));
    {
        std::vector<float> hr(rin_n);
        for (auto& v : hr) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        CK(hipMemcpy(rin, hr.data(), rin_n * 4, hipMemcpyHostToDevice));
    }
    for (int co = 0; co < 2; ++co) {   // the compiled resize beside co-runner 1 (registers only) and 2 (global -> LDS -> MFMA)
        hipLaunchKernelGGL(resize_kernel, dim3(1024), dim3(256), 0, s1, rin, rref, RB, RH, RW, RO, RO, RC);
        CK(hipStreamSynchronize(s1));
        std::vector<float> a0(rout_n), a1(rout_n);
        CK(hipMemcpy(a0.data(), rref, rout_n * 4, hipMemcpyDeviceToHost));
        long diff = 0;
        for (int r = 0; r < rounds; ++r) {
            if (co == 0) hipLaunchKernelGGL(mfma_corunner, dim3(1024), dim3(256), 0, s2, sink, 400000);
            else hipLaunchKernelGGL(gemm_like_corunner, dim3(1024), dim3(256), 0, s2, gsrc, sink, 60000, nsrc);
            CK(hipMemsetAsync(rout, 0xff, rout_n * 4, s1));
            for (int k = 0; k < 20; ++k)   // (the resize is short: several launches fall inside one co-runner)
                hipLaunchKernelGGL(resize_kernel, dim3(1024), dim3(256), 0, s1, rin, rout, RB, RH, RW, RO, RO, RC);
            CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(a1.data(), rout, rout_n * 4, hipMemcpyDeviceToHost));
            CK(hipStreamSynchronize(s2));
            for (size_t i = 0; i < rout_n; ++i) diff += memcmp(&a0[i], &a1[i], 4) != 0;
        }
        printf("compiled resize (hipcc -O3, SLP on) beside %s: %ld differing floats over %d rounds of %zu\n",
               co == 0 ? "the register-only bf16 MFMA loop" : "the global -> LDS -> bf16 MFMA co-runner", diff, rounds, rout_n);
    }
    std::vector<f2> href(n), hout(n);
    long total_diff[2] = {0, 0}, packed_vs_scalar = 0;
    for (int arm = 0; arm < 2; ++arm) {  // 0 = packed, 1 = scalar twin
        auto launch = [&](f2* o) {
            if (arm == 0) hipLaunchKernelGGL(packed_kernel, dim3(2048), dim3(256), 0, s1, a, b, c, d, w, o, n, reps);
            else hipLaunchKernelGGL(scalar_kernel, dim3(2048), dim3(256), 0, s1, a, b, c, d, w, o, n, reps);
        };
        launch(ref);  // alone on the chip
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(href.data(), ref, n * sizeof(f2), hipMemcpyDeviceToHost));
        if (arm == 0) hout = href;  // packed result, kept for the packed-vs-scalar comparison below
        else {
            for (int i = 0; i < n; ++i) packed_vs_scalar += memcmp(&hout[i], &href[i], sizeof(f2)) != 0;
        }
        std::vector<f2> got(n);
        for (int r = 0; r < rounds; ++r) {
            // a co-runner of ~30 ms on every CU (4 workgroups of 4 waves per CU), the kernel under test launched into the middle of it
            hipLaunchKernelGGL(mfma_corunner, dim3(1024), dim3(256), 0, s2, sink, 400000);
            CK(hipMemsetAsync(out, 0xff, n * sizeof(f2), s1));
            launch(out);
            CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(got.data(), out, n * sizeof(f2), hipMemcpyDeviceToHost));
            CK(hipStreamSynchronize(s2));
            long diff = 0;
            for (int i = 0; i < n; ++i) diff += memcmp(&got[i], &href[i], sizeof(f2)) != 0;
            total_diff[arm] += diff;
        }
        printf("%s kernel: %ld differing float2 over %d co-run rounds of %d elements (0 = bit-identical to the run alone)\n",
               arm == 0 ? "packed (v_pk_*_f32 with op_sel)" : "scalar twin (v_mul_f32 / v_fma_f32)", total_diff[arm], rounds, n);
    }
    printf("packed vs scalar, both alone: %ld differing float2 (0 = the two forms are the same arithmetic)\n", packed_vs_scalar);
    printf("RESULT: %s\n", total_diff[0] != 0 && total_diff[1] == 0
                               ? "REPRODUCED - packed fp32 with op_sel differs beside bf16 MFMA waves, the scalar twin does not"
                               : (total_diff[0] == 0 && total_diff[1] == 0 ? "NOT reproduced - both forms are bit-stable beside the bf16 MFMA co-runner"
                                                                            : "inconclusive - the scalar twin differs too"));
    return 0;
}
