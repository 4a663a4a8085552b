// Do memory operations between MFMA phases slow the MFMA phases of the OTHER waves of a CU?  Each wave: `tiles` x { 4 * nm
// register-only MFMAs, then four 16-byte-per-lane memory instructions }.  MODE 1 = MFMAs only, 2 = memory only, 3 = both.
// PAT 0: stores in the GEMM epilogue's pattern (16 rows x 64 B per wave-instruction), streaming (302 MB to HBM);
// PAT 1: the same stores, every tile of a block onto the same 16 KB (L2 hits, no HBM traffic); PAT 2: loads instead of stores;
// PAT 3: non-temporal stores; PAT 4: stores cycling within 32 MB (tile index modulo 1); PAT 5: 1 KiB contiguous per instruction.
// 2048 workgroups of 4 waves = 8 waves per SIMD, all resident: the residency of the 64x64 GEMM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int PAT, int STAG = 0>
__global__ __launch_bounds__(256) void k(float* out, int tiles, int nm, float a0, long stride_tile, unsigned long long* clk = nullptr) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = a0 + threadIdx.x * 1e-3f, b = a0 * 0.5f + threadIdx.x * 2e-3f;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, i16 = lane & 15;
    float extra = 0.f;
    // STAG: the eight co-resident workgroups of a CU start 1/8 of a tile time apart (all waves otherwise march in lockstep:
    // everybody multiplies, then everybody touches memory)
    if (STAG) for (int d = 0; d < (int)((blockIdx.x >> 8) & 7) * nm / 8; ++d) __builtin_amdgcn_s_sleep(16);
    for (int t = 0; t < tiles; ++t) {
        if (MODE & 1)
            for (int it = 0; it < nm; ++it)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (MODE & 2) {
            // PAT 4 / 6 / 7: a block cycles over 2 / 4 / 1 tiles: 64 / 128 / 32 MB footprint (MALL-resident, not L2-resident)
            float* base = out + ((long)blockIdx.x * tiles + (PAT == 1 ? 0 : PAT == 4 ? t % 2 : PAT == 6 ? t % 4 : PAT == 7 ? 0 : t)) * stride_tile + (PAT == 7 ? (long)(t % 1) : 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (wid >> 1) * 32 + (i >> 1) * 16 + i16, n = (wid & 1) * 32 + (i & 1) * 16 + 4 * g;
                if (PAT == 2) { const float4 r = *reinterpret_cast<const float4*>(base + m * 64 + n); extra += r.x; }
                else if (PAT == 3) __builtin_nontemporal_store(f32x4{a + t, b, a, b + i}, reinterpret_cast<f32x4*>(base + m * 64 + n));
                else if (PAT >= 10) {  // cache-policy bits on the store: 10 = sc0, 11 = sc1, 12 = sc0 sc1, 13 = sc0 sc1 nt
                    const f32x4 v = {a + t, b, a, b + i};
                    float* dst = base + m * 64 + n;
                    if (PAT == 10) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(dst), "v"(v) : "memory");
                    if (PAT == 11) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
                    if (PAT == 12) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
                    if (PAT == 13) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(dst), "v"(v) : "memory");
                }
                else if (PAT == 5) *reinterpret_cast<float4*>(base + (wid * 4 + i) * 256 + lane * 4) = make_float4(a + t, b, a, b + i);
                else *reinterpret_cast<float4*>(base + m * 64 + n) = make_float4(a + t, b, a, b + i);
            }
        }
    }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + extra == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = 1.f;  // keeps the MFMAs live
    if (clk && threadIdx.x == 0 && blockIdx.x == 777) {  // shader clock: cycle counter against the 100 MHz real-time counter
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
template <int PAT>
void run(float* out, int blocks, int tiles, long stride_tile, hipEvent_t e0, hipEvent_t e1) {
    for (int nm : {24}) {
        float ms[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); hipLaunchKernelGGL((k<1, PAT>), dim3(blocks), dim3(256), 0, 0, out, tiles, nm, 0.5f, stride_tile); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[1], e0, e1);
            (void)hipEventRecord(e0); hipLaunchKernelGGL((k<2, PAT>), dim3(blocks), dim3(256), 0, 0, out, tiles, nm, 0.5f, stride_tile); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[2], e0, e1);
            (void)hipEventRecord(e0); hipLaunchKernelGGL((k<3, PAT>), dim3(blocks), dim3(256), 0, 0, out, tiles, nm, 0.5f, stride_tile); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[3], e0, e1);
            (void)hipEventRecord(e0); hipLaunchKernelGGL((k<3, PAT, 1>), dim3(blocks), dim3(256), 0, 0, out, tiles, nm, 0.5f, stride_tile); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[0], e0, e1);
        }
        printf("pattern %d, %3d MFMAs per wave-tile: mfma only %.1f us, memory only %.1f us, both %.1f us, both staggered %.1f us  (sum %.1f, max %.1f)\n", PAT, 4 * nm,
               ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3, ms[0] * 1e3, (ms[1] + ms[2]) * 1e3, (ms[1] > ms[2] ? ms[1] : ms[2]) * 1e3);
    }
}
template <int MODE, int PAT, int STAG>
float timed(float* out, int blocks, int tiles, int nm, long stride_tile, hipEvent_t e0, hipEvent_t e1) {
    float ms;
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, PAT, STAG>), dim3(blocks), dim3(256), 0, 0, out, tiles, nm, 0.5f, stride_tile, nullptr); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    const int blocks = 2048, tiles = 9, nm = 24;
    const long stride_tile = 64 * 64;
    float* out; (void)hipMalloc(&out, (size_t)blocks * tiles * stride_tile * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // what does the time of "MFMAs + streaming stores" depend on?  the kernel that ran before it:
    for (int rep = 0; rep < 2; ++rep) {
        float a = timed<1, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float b = timed<3, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float c = timed<3, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float d = timed<3, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float e = timed<2, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float f = timed<3, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float g = timed<1, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float h = timed<3, 0, 1>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float i = timed<3, 0, 1>(out, blocks, tiles, nm, stride_tile, e0, e1);
        float j = timed<3, 0, 0>(out, blocks, tiles, nm, stride_tile, e0, e1);
        printf("mfma %.0f | both %.0f, both %.0f, both %.0f | stores %.0f | both %.0f | mfma %.0f | both-stag %.0f, both-stag %.0f | both %.0f\n", a, b, c, d, e, f, g, h, i, j);
    }
    return 0;
}
