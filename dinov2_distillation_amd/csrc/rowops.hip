// Row-wise and column-wise HBM-bound kernels: LayerNorm fwd/bwd (one wavefront per token row, row held in
// registers, wave-shuffle reductions), grouped column reductions (bias/affine gradients, BatchNorm statistics,
// GRN energies, position-table gradients) and a tiled transpose.  All loads/stores are 16 B per lane.
#include "common.h"
#include "dkd_hip.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane -> C <= 2048

template <bool WRITE_STATS>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long rows, int C, float eps, int in_grp, int in_grp_stride,
                                                            int in_off) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = C >> 2;
    const float invC = 1.f / (float)C;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        const long irow = in_grp > 0 ? (row / in_grp) * in_grp_stride + row % in_grp + in_off : row;
        const float4* xr = reinterpret_cast<const float4*>(x + irow * C);
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(wave_sum(q) * invC + eps);
        float4* yr = reinterpret_cast<float4*>(y + row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                const float4 b = reinterpret_cast<const float4*>(beta)[c];
                float4 o;
                o.x = (v[i].x - mu) * rs * g.x + b.x;
                o.y = (v[i].y - mu) * rs * g.y + b.y;
                o.z = (v[i].z - mu) * rs * g.z + b.z;
                o.w = (v[i].w - mu) * rs * g.w + b.w;
                yr[c] = o;
            }
        }
        if (WRITE_STATS && lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  Per-block partial sums of
// dgamma = sum dy*xhat and dbeta = sum dy go to ws[block][2][C] (AFFINE) and are folded by reduce_partials.
template <bool AFFINE>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ add, float* __restrict__ dx,
                                                            float* __restrict__ ws, long rows, int C) {
    __shared__ float red[AFFINE ? 4 * LN_MAXV * 256 : 4];  // [wave][C <= 2048]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = C >> 2;
    const float invC = 1.f / (float)C;
    float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * C);
        const float4* dr = reinterpret_cast<const float4*>(dy + row * C);
        const float mu = mean[row], rs = rstd[row];
        float4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float4 xv = xr[c], dv = dr[c], gm = reinterpret_cast<const float4*>(gamma)[c];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                if (AFFINE) {
                    ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
                    ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
                }
            } else {
                xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
        float4* oxr = reinterpret_cast<float4*>(dx + row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                float4 o;
                o.x = rs * (g[i].x - s1 - xh[i].x * s2);
                o.y = rs * (g[i].y - s1 - xh[i].y * s2);
                o.z = rs * (g[i].z - s1 - xh[i].z * s2);
                o.w = rs * (g[i].w - s1 - xh[i].w * s2);
                if (add) {
                    const float4 a = reinterpret_cast<const float4*>(add + row * C)[c];
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                oxr[c] = o;
            }
        }
    }
    if (AFFINE) {
        // fold the 4 waves of the block through LDS, then one partial row per block
        float* r = red;  // dgamma and dbeta are folded one after the other through the same 32 KiB
        for (int part = 0; part < 2; ++part) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c = lane + i * 64;
                if (c < nv) reinterpret_cast<float4*>(r + wid * (LN_MAXV * 256))[c] = part == 0 ? ag[i] : ab[i];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += 256) {
                const float s = (r[c] + r[LN_MAXV * 256 + c]) + (r[2 * LN_MAXV * 256 + c] + r[3 * LN_MAXV * 256 + c]);
                ws[((long)blockIdx.x * 2 + part) * C + c] = s;
            }
        }
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nparts, int n,
                                                              float* __restrict__ out0, float* __restrict__ out1, int split,
                                                              float scale) {
    // out[j] = scale * sum_p ws[p][j]; columns [0,split) go to out0, [split,n) to out1.  64 columns x 4 part-lanes.
    __shared__ float red[256];
    const int c = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s = 0.f;
    if (j < n)
        for (int p = pl; p < nparts; p += 4) s += ws[(long)p * n + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && j < n) {
        s = ((red[c] + red[64 + c]) + (red[128 + c] + red[192 + c])) * scale;
        if (j < split) out0[j] = s;
        else out1[j - split] = s;
    }
}

// ---- grouped column reduction -----------------------------------------------------------------------------
// out[g][n] = scale * sum_{r < rows_per_group} f(X[(g*rows_per_group + r)*ldx + n], Y[...], aux[n])
template <int OP>
__device__ __forceinline__ float4 colop(const float4 x, const float4 y, const float4 a) {
    if (OP == DKD_RED_SUM) return x;
    if (OP == DKD_RED_DOT) return make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    if (OP == DKD_RED_SUMSQ) return make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w);
    // DKD_RED_SQDIFF: (x - aux)^2
    const float4 d = make_float4(x.x - a.x, x.y - a.y, x.z - a.z, x.w - a.w);
    return make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w);
}

// OP == DKD_RED_DOT_SUM: two results per column, sum x*y -> out[.], sum x -> out[groups*nsplit*N + .]
template <int OP>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                        const float* __restrict__ aux, float* __restrict__ out,
                                                        long rows_per_group, int N, long ldx, int CL, int rows_per_split,
                                                        float scale, int nsplit) {
    __shared__ float4 red[256];
    const int RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c4 = blockIdx.x * CL + cl;  // float4 column index
    const int g = blockIdx.z, sp = blockIdx.y;
    const long r0 = (long)sp * rows_per_split;
    const long r1 = min(rows_per_group, r0 + rows_per_split);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
    constexpr bool TWO = OP == DKD_RED_DOT_SUM;
    if (4 * c4 < N) {
        const float4 a = OP == DKD_RED_SQDIFF ? reinterpret_cast<const float4*>(aux)[c4] : acc;
        const long base = (long)g * rows_per_group;
        for (long r = r0 + rl; r < r1; r += RL) {
            const float4 x = *reinterpret_cast<const float4*>(X + (base + r) * ldx + 4 * c4);
            float4 y = x;
            if (OP == DKD_RED_DOT || TWO) y = *reinterpret_cast<const float4*>(Y + (base + r) * ldx + 4 * c4);
            const float4 v = colop<TWO ? DKD_RED_DOT : OP>(x, y, a);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            if (TWO) { acc2.x += x.x; acc2.y += x.y; acc2.z += x.z; acc2.w += x.w; }
        }
    }
    const long second = (long)gridDim.z * nsplit * N;  // offset of the second result plane
    for (int pass = 0; pass < (TWO ? 2 : 1); ++pass) {
        if (pass == 1) { __syncthreads(); acc = acc2; }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0 && 4 * c4 < N) {
            for (int k = 1; k < RL; ++k) {
                const float4 v = red[k * CL + cl];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            // partial layout [g][split][N]; with nsplit == 1 this IS the output
            float4* o = reinterpret_cast<float4*>(out + pass * second + ((long)g * nsplit + sp) * N) + c4;
            if (nsplit == 1) { acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale; }
            *o = acc;
        }
    }
}

__global__ void colreduce_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int groups, int nsplit,
                                       int N, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)groups * N) return;
    const int g = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int p = 0; p < nsplit; ++p) s += ws[((long)g * nsplit + p) * N + n];
    out[i] = s * scale;
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int C, long in_bs, long out_bs) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    in += (long)blockIdx.z * in_bs;
    out += (long)blockIdx.z * out_bs;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (by + j < R && bx + tx < C) tile[j][tx] = in[(long)(by + j) * C + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < C && by + tx < R) out[(long)(bx + j) * R + by + tx] = tile[tx][j];
}

void colreduce_plan(long rows_per_group, int groups, int N, int* CL, int* nsplit, int* rows_per_split, int* colblocks) {
    const int nv = N / 4;
    int cl = 1;
    while (cl < nv && cl < 64) cl <<= 1;
    const int RL = 256 / cl;
    const int cb = (nv + cl - 1) / cl;
    long want = 1024 / ((long)cb * groups);  // ~4 blocks per CU in total
    if (want < 1) want = 1;
    long maxs = rows_per_group / ((long)RL * 8);  // >= 8 rows per thread
    if (maxs < 1) maxs = 1;
    long ns = want < maxs ? want : maxs;
    if (ns > 64) ns = 64;  // the second pass walks the partials serially per column: keep it short
    long rps = (rows_per_group + ns - 1) / ns;
    ns = (rows_per_group + rps - 1) / rps;
    *CL = cl; *nsplit = (int)ns; *rows_per_split = (int)rps; *colblocks = cb;
}

}  // namespace

extern "C" int dkd_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                     float* rstd, long rows, int C, float eps, int in_row_grp, int in_row_grp_stride,
                                     int in_row_off, void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int blocks = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    if (mean && rstd)
        hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, rows, C, eps, in_row_grp, in_row_grp_stride, in_row_off);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta,
                           y, mean, rstd, rows, C, eps, in_row_grp, in_row_grp_stride, in_row_off);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

static int ln_bwd_blocks(long rows) { return (int)((rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256); }

extern "C" long dkd_layernorm_bwd_f32_workspace_bytes(long rows, int C) {
    return (long)ln_bwd_blocks(rows) * 2 * C * sizeof(float);
}

extern "C" int dkd_layernorm_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                     float* ws, long rows, int C, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || rows <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int blocks = ln_bwd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    if (dgamma && dbeta) {
        if (!ws) return DKD_ERR_BAD_ARG;
        hipLaunchKernelGGL(layernorm_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, dy, x, mean, rstd, gamma, add, dx,
                           ws, rows, C);
        DKD_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, st, ws, blocks, 2 * C, dgamma,
                           dbeta, C, 1.0f);
    } else {
        hipLaunchKernelGGL(layernorm_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, dy, x, mean, rstd, gamma, add, dx,
                           ws, rows, C);
    }
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" long dkd_colreduce_f32_workspace_bytes(long rows_per_group, int groups, int N) {
    int cl, ns, rps, cb;
    colreduce_plan(rows_per_group, groups, N, &cl, &ns, &rps, &cb);
    return ns > 1 ? 2L * groups * ns * N * sizeof(float) : 0;  // x2: DKD_RED_DOT_SUM keeps two planes
}

extern "C" int dkd_colreduce_f32(const float* X, const float* Y, const float* aux, float* out, float* ws,
                                 long rows_per_group, int groups, int N, long ldx, int op, float scale, void* stream) {
    if (!X || !out || rows_per_group <= 0 || groups <= 0 || N <= 0) return DKD_ERR_BAD_ARG;
    if ((N & 3) || (ldx & 3)) return DKD_ERR_ALIGN;
    if ((op == DKD_RED_DOT || op == DKD_RED_DOT_SUM) && !Y) return DKD_ERR_BAD_ARG;
    if (op == DKD_RED_SQDIFF && !aux) return DKD_ERR_BAD_ARG;
    int cl, ns, rps, cb;
    colreduce_plan(rows_per_group, groups, N, &cl, &ns, &rps, &cb);
    if (ns > 1 && !ws) return DKD_ERR_BAD_ARG;
    float* dst = ns > 1 ? ws : out;
    dim3 grid(cb, ns, groups), block(256);
    hipStream_t st = (hipStream_t)stream;
#define DKD_CR(OPV)                                                                                               \
    hipLaunchKernelGGL(colreduce_kernel<OPV>, grid, block, 0, st, X, Y, aux, dst, rows_per_group, N, ldx, cl, rps, \
                       scale, ns)
    switch (op) {
        case DKD_RED_SUM: DKD_CR(DKD_RED_SUM); break;
        case DKD_RED_DOT: DKD_CR(DKD_RED_DOT); break;
        case DKD_RED_SUMSQ: DKD_CR(DKD_RED_SUMSQ); break;
        case DKD_RED_SQDIFF: DKD_CR(DKD_RED_SQDIFF); break;
        case DKD_RED_DOT_SUM: DKD_CR(DKD_RED_DOT_SUM); break;
        default: return DKD_ERR_BAD_ARG;
    }
#undef DKD_CR
    DKD_LAUNCH_CHECK();
    if (ns > 1) {
        const long total = (long)groups * N;
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, ws, out, groups, ns,
                           N, scale);
        DKD_LAUNCH_CHECK();
        if (op == DKD_RED_DOT_SUM) {
            hipLaunchKernelGGL(colreduce_final_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st,
                               ws + (long)groups * ns * N, out + total, groups, ns, N, scale);
            DKD_LAUNCH_CHECK();
        }
    }
    return DKD_OK;
}

extern "C" int dkd_transpose_f32(const float* in, float* out, int rows, int cols, int batch, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || batch <= 0) return DKD_ERR_BAD_ARG;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, rows, cols,
                       (long)rows * cols, (long)rows * cols);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
