// Row-wise and column-wise HBM-bound kernels: LayerNorm fwd/bwd (one wavefront per token row, row held in
// registers, wave-shuffle reductions), grouped column reductions (bias/affine gradients, BatchNorm statistics,
// GRN energies, position-table gradients) and a tiled transpose.  All loads/stores are 16 B per lane.
#include "common.h"
#include "dkd_hip.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane -> C <= 2048

// Rows narrower than a wave share it: LPR lanes per row (16 / 32 / 64), 64/LPR rows per wave, VPL float4 per lane
// (C <= 4 * LPR * VPL).  ConvNeXt's C = 96 rows (24 float4) used to occupy 24 of 64 lanes; now two rows ride one wave.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));

// OUT_BF16: y is bf16 (the autocast teacher: LayerNorm computes in fp32 and its only consumer, an nn.Linear, rounds its input
// to 16 bit - the rounding is done here and the [rows, C] tensor costs half the bytes)
// CF16 (bf16 input only): the reference's channels-first LayerNorm (convnext.py:31-36) as torch autocast runs it op by op on a
// 16-bit tensor: u = x.mean() and d = x - u are 16-bit results (rounded), pow / mean / sqrt / the affine part fp32.  `mean` then
// holds the rounded u and the variance is that of the rounded d about zero.
template <int LPR, int VPL, bool WRITE_STATS, bool OUT_BF16 = false, typename TI = float, bool CF16 = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long rows, int C, float eps, int in_grp, int in_grp_stride,
                                                            int in_off) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane / LPR, lir = lane % LPR;
    const int nv = C >> 2;
    const float invC = 1.f / (float)C;
    float4 gm[VPL], bt[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lir + i * LPR;
        gm[i] = c < nv ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[i] = c < nv ? reinterpret_cast<const float4*>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = ((long)blockIdx.x * 4 + wid) * RPW + sub; row < rows; row += (long)gridDim.x * 4 * RPW) {
        const long irow = in_grp > 0 ? (row / in_grp) * in_grp_stride + row % in_grp + in_off : row;
        const long xr = irow * nv;  // 4-element index of the row
        float4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lir + i * LPR;
            v[i] = c < nv ? ld4(x, xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        float mu = group_sum<LPR>(s) * invC;
        if (CF16) mu = rnd16(mu);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lir + i * LPR;
            if (c < nv) {
                float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                if (CF16) {  // keep the rounded differences: the output below is built from them
                    a = rnd16(a), b = rnd16(b), cc = rnd16(cc), d = rnd16(d);
                    v[i] = make_float4(a, b, cc, d);
                }
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(group_sum<LPR>(q) * invC + eps);
        float4* yr = reinterpret_cast<float4*>(y + row * C);
        ln_bf16x4* yb = reinterpret_cast<ln_bf16x4*>(reinterpret_cast<__bf16*>(y) + row * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lir + i * LPR;
            if (c < nv) {
                const float sub = CF16 ? 0.f : mu;
                float4 o;
                o.x = (v[i].x - sub) * rs * gm[i].x + bt[i].x;
                o.y = (v[i].y - sub) * rs * gm[i].y + bt[i].y;
                o.z = (v[i].z - sub) * rs * gm[i].z + bt[i].z;
                o.w = (v[i].w - sub) * rs * gm[i].w + bt[i].w;
                if (OUT_BF16) yb[c] = ln_bf16x4{(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
                else yr[c] = o;
            }
        }
        if (WRITE_STATS && lir == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  Per-block partial sums of
// dgamma = sum dy*xhat and dbeta = sum dy go to ws[block][2][C] (AFFINE) and are folded by reduce_partials.
// CF16 (bf16 x): the adjoint of the CF16 forward as autograd forms it - d's two 16-bit gradients (through d / sqrt(s) and through
// the variance) are added and rounded, the mean's gradient is the rounded negative channel sum, and x receives
// rnd(dd + rnd(du / C)); the statistics are those of the rounded d (xhat = rnd(x - u) * rstd)
template <int LPR, int VPL, bool AFFINE, typename TD = float, typename TI = float, typename TO = float, bool CF16 = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TD* __restrict__ dy, const TI* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ add, TO* __restrict__ dx,
                                                            float* __restrict__ ws, long rows, int C) {
    constexpr int RPW = 64 / LPR, SLOTS = 4 * RPW, CMAX = 4 * LPR * VPL;
    __shared__ float red[AFFINE ? SLOTS * CMAX : 4];  // [wave x sub-row][C]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sub = lane / LPR, lir = lane % LPR;
    const int nv = C >> 2;
    const float invC = 1.f / (float)C;
    float4 ag[VPL], ab[VPL], gm[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lir + i * LPR;
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[i] = c < nv ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = ((long)blockIdx.x * 4 + wid) * RPW + sub; row < rows; row += (long)gridDim.x * 4 * RPW) {
        const long r4 = row * nv;  // 4-element index of the row
        const float mu = mean[row], rs = rstd[row];
        float4 xh[VPL], g[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lir + i * LPR;
            if (c < nv) {
                const float4 xv = ld4(x, r4 + c), dv = ld4(dy, r4 + c);
                xh[i] = CF16 ? make_float4(rnd16(xv.x - mu) * rs, rnd16(xv.y - mu) * rs, rnd16(xv.z - mu) * rs, rnd16(xv.w - mu) * rs)
                             : make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm[i].x, dv.y * gm[i].y, dv.z * gm[i].z, dv.w * gm[i].w);
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
        s1 = group_sum<LPR>(s1) * invC;
        s2 = group_sum<LPR>(s2) * invC;
        if (CF16) {
            float4 dd[VPL];
            float sd = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = lir + i * LPR;
                dd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < nv) {
                    const float4 t1 = rnd16(make_float4(g[i].x * rs, g[i].y * rs, g[i].z * rs, g[i].w * rs));
                    const float4 t2 = rnd16(make_float4(-xh[i].x * s2 * rs, -xh[i].y * s2 * rs, -xh[i].z * s2 * rs, -xh[i].w * s2 * rs));
                    dd[i] = rnd16(make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w));
                    sd += (dd[i].x + dd[i].y) + (dd[i].z + dd[i].w);
                }
            }
            const float du = rnd16(rnd16(-group_sum<LPR>(sd)) * invC);
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = lir + i * LPR;
                if (c < nv) st4(dx, r4 + c, make_float4(dd[i].x + du, dd[i].y + du, dd[i].z + du, dd[i].w + du));
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lir + i * LPR;
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
                st4(dx, r4 + c, o);
            }
        }
    }
    if (AFFINE) {
        // fold the (4 waves x RPW sub-rows) accumulators of the block through LDS, then one partial row per block
        const int slot = wid * RPW + sub;
        for (int part = 0; part < 2; ++part) {  // dgamma and dbeta one after the other through the same buffer
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = lir + i * LPR;
                if (c < nv) reinterpret_cast<float4*>(red + slot * CMAX)[c] = part == 0 ? ag[i] : ab[i];
            }
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += 256) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < SLOTS; ++k) s += red[k * CMAX + c];
                ws[((long)blockIdx.x * 2 + part) * C + c] = s;
            }
        }
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nparts, int n,
                                                              float* __restrict__ out0, float* __restrict__ out1, int split,
                                                              float scale) {
    // out[j] = scale * sum_p ws[p][j]; columns [0,split) go to out0, [split,n) to out1 (n, split % 4 == 0).
    // 16 float4 column lanes x 16 part lanes per block; each thread keeps four partial rows in flight.
    __shared__ float4 red[256];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int j4 = blockIdx.x * 16 + cl, n4 = n >> 2;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j4 < n4) {
        const float4* w = reinterpret_cast<const float4*>(ws) + j4;
        int p = pl;
        for (; p + 48 < nparts; p += 64) {
            const float4 a = w[(long)p * n4], b = w[(long)(p + 16) * n4], c = w[(long)(p + 32) * n4], d = w[(long)(p + 48) * n4];
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
            s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; p < nparts; p += 16) {
            const float4 a = w[(long)p * n4];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && j4 < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float4 a = red[k * 16 + cl];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
        const int j = j4 << 2;
        if (j < split) *reinterpret_cast<float4*>(out0 + j) = s;
        else *reinterpret_cast<float4*>(out1 + (j - split)) = s;
    }
}

// s += v with the rounding error of the addition carried in c (Kahan); the caller finishes with s -= c
__device__ __forceinline__ void kahan1(float& s, float& c, const float v) {
    const float y = v - c;
    const float t = s + y;
    c = (t - s) - y;
    s = t;
}
__device__ __forceinline__ void kahan4(float4& s, float4& c, const float4 v) {
    kahan1(s.x, c.x, v.x); kahan1(s.y, c.y, v.y); kahan1(s.z, c.z, v.z); kahan1(s.w, c.w, v.w);
}

// ---- grouped column reduction -----------------------------------------------------------------------------
// out[g][n] = scale * sum_{r < rows_per_group} f(X[(g*rows_per_group + r)*ldx + n], Y[...], aux[n])
template <int OP>
__device__ __forceinline__ float4 colop(const float4 x, const float4 y, const float4 a) {
    if (OP == DKD_RED_SUM) return x;
    if (OP == DKD_RED_DOT) return make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
    if (OP == DKD_RED_SUMSQ) return make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w);
    if (OP == DKD_RED_GELUSQ) {
        const float4 g = make_float4(gelu_f(x.x), gelu_f(x.y), gelu_f(x.z), gelu_f(x.w));
        return make_float4(g.x * g.x, g.y * g.y, g.z * g.z, g.w * g.w);
    }
    // DKD_RED_SQDIFF: (x - aux)^2
    const float4 d = make_float4(x.x - a.x, x.y - a.y, x.z - a.z, x.w - a.w);
    return make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w);
}

// OP == DKD_RED_DOT_SUM: two results per column, sum x*y -> out[.], sum x -> out[groups*nsplit*N + .]
// T: element type of X and Y (float or __bf16).  R16 (the mixed-precision step): the GELU inside the gelu variants is rounded to
// bf16 first - the value the 16-bit GELU output tensor of the autocast step holds.
// OP == DKD_RED_BN_BWD (X = the gradient behind BatchNorm + activation, Y = the BatchNorm INPUT z, bn = the per-column mean / rstd /
// gamma / beta): the two sums of BatchNorm's backward pass in one pass over the two tensors, dbn = X * act'(bn(z)) and
// xhat = (z - mean) * rstd recomputed on the fly: plane 0 = sum dbn * xhat (d gamma), plane 1 = sum dbn (d beta).
template <int OP, typename T = float, bool R16 = false>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ X, const T* __restrict__ Y,
                                                        const float* __restrict__ aux, float* __restrict__ out,
                                                        long rows_per_group, int N, long ldx, int CL, int rows_per_split,
                                                        float scale, int nsplit, int round_out = 0, dkd_bn_cols bn = dkd_bn_cols{}) {
    __shared__ float4 red[256];
    const int RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c4 = blockIdx.x * CL + cl;  // float4 column index
    const int g = blockIdx.z, sp = blockIdx.y;
    const long r0 = (long)sp * rows_per_split;
    const long r1 = min(rows_per_group, r0 + rows_per_split);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc, cmp = acc, cmp2 = acc;
    constexpr bool TWO = OP == DKD_RED_DOT_SUM || OP == DKD_RED_DOTGELU_SUM || OP == DKD_RED_BN_BWD;
    if (4 * c4 < N) {
        const float4 a = OP == DKD_RED_SQDIFF ? reinterpret_cast<const float4*>(aux)[c4] : acc;
        float4 mu = acc, rs = acc, gm = acc, bt = acc;
        if (OP == DKD_RED_BN_BWD) {
            mu = reinterpret_cast<const float4*>(bn.mean)[c4], rs = reinterpret_cast<const float4*>(bn.rstd)[c4];
            gm = reinterpret_cast<const float4*>(bn.gamma)[c4], bt = reinterpret_cast<const float4*>(bn.beta)[c4];
        }
        const long base = (long)g * rows_per_group;
        // Kahan-compensated accumulation: a column of B*H*W gradients summed in plain fp32 carries an error that grows with the
        // number of rows a thread folds; the BatchNorm / bias gradients of a 50-layer student amplify it (the PResNet-50 case of
        // round 1).  The kernel is HBM-bound, the three extra VALU ops per element are free (scripts/bench_colreduce.py).
        auto fold = [&](float4 x, float4 y) {
            if (OP == DKD_RED_BN_BWD) {
                const float4 xh = make_float4((y.x - mu.x) * rs.x, (y.y - mu.y) * rs.y, (y.z - mu.z) * rs.z, (y.w - mu.w) * rs.w);
                if (bn.act) {
                    x.x = bn_act_grad(x.x, gm.x * xh.x + bt.x, bn.act); x.y = bn_act_grad(x.y, gm.y * xh.y + bt.y, bn.act);
                    x.z = bn_act_grad(x.z, gm.z * xh.z + bt.z, bn.act); x.w = bn_act_grad(x.w, gm.w * xh.w + bt.w, bn.act);
                }
                kahan4(acc, cmp, make_float4(x.x * xh.x, x.y * xh.y, x.z * xh.z, x.w * xh.w));
                kahan4(acc2, cmp2, x);
                return;
            }
            if (OP == DKD_RED_DOTGELU_SUM) {
                y = make_float4(gelu_f(y.x), gelu_f(y.y), gelu_f(y.z), gelu_f(y.w));
                if (R16) y = rnd16(y);
            }
            if (OP == DKD_RED_GELUSQ && R16) {  // (gelu rounded, then squared: colop's own gelu is bypassed)
                const float4 g = rnd16(make_float4(gelu_f(x.x), gelu_f(x.y), gelu_f(x.z), gelu_f(x.w)));
                kahan4(acc, cmp, make_float4(g.x * g.x, g.y * g.y, g.z * g.z, g.w * g.w));
                return;
            }
            const float4 v = colop<TWO ? DKD_RED_DOT : OP>(x, y, a);
            kahan4(acc, cmp, v);
            if (TWO) kahan4(acc2, cmp2, x);
        };
        constexpr bool HAS_Y = OP == DKD_RED_DOT || TWO;  // (DKD_RED_BN_BWD included)
        long r = r0 + rl;
        // keep several rows' loads in flight ahead of the (for the gelu variants VALU-heavy) arithmetic: four rows for the
        // one-operand reductions, the plain loop for the two-operand ones (eight float4 in flight cost occupancy: +12 %)
        if (!HAS_Y) {
            const long l4 = ldx >> 2;
            for (; r + 3 * RL < r1; r += 4 * RL) {
                const long px = (base + r) * l4 + c4;  // 4-element index
                const float4 x0 = ld4(X, px), x1 = ld4(X, px + RL * l4);
                const float4 x2 = ld4(X, px + 2 * RL * l4), x3 = ld4(X, px + 3 * RL * l4);
                fold(x0, x0); fold(x1, x1); fold(x2, x2); fold(x3, x3);
            }
        }
        for (; r < r1; r += RL) {
            const float4 x = ld4(X, (base + r) * (ldx >> 2) + c4);
            float4 y = x;
            if (HAS_Y) y = ld4(Y, (base + r) * (ldx >> 2) + c4);
            fold(x, y);
        }
    }
    const long second = (long)gridDim.z * nsplit * N;  // offset of the second result plane
    for (int pass = 0; pass < (TWO ? 2 : 1); ++pass) {
        if (pass == 1) { __syncthreads(); acc = acc2; cmp = cmp2; }
        acc.x -= cmp.x; acc.y -= cmp.y; acc.z -= cmp.z; acc.w -= cmp.w;
        red[threadIdx.x] = acc;
        __syncthreads();
        if (rl == 0 && 4 * c4 < N) {
            cmp = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 1; k < RL; ++k) kahan4(acc, cmp, red[k * CL + cl]);
            acc.x -= cmp.x; acc.y -= cmp.y; acc.z -= cmp.z; acc.w -= cmp.w;
            // partial layout [g][split][N]; with nsplit == 1 this IS the output
            float4* o = reinterpret_cast<float4*>(out + pass * second + ((long)g * nsplit + sp) * N) + c4;
            if (OP == DKD_RED_BN_BWD && pass == 1 && nsplit == 1 && bn.out2) o = reinterpret_cast<float4*>(bn.out2) + c4;
            if (nsplit == 1) {
                acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
                if (round_out) acc = rnd16(acc);  // (a 16-bit reduction result held in fp32: the mixed-precision step's bias gradients)
            }
            *o = acc;
        }
    }
}

// out[g][n] = scale * sum_p ws[g][p][n]: 32 column lanes x 8 split lanes per block (grid: column chunks x groups)
// (grid.z = result planes: the two planes of DKD_RED_DOT_SUM / DOTGELU_SUM are folded by ONE launch)
// tr_T > 0: the result, read as [tr_T][N / tr_T], is stored transposed: out[(n % (N / tr_T)) * tr_T + n / (N / tr_T)]
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int groups,
                                                              int nsplit, int N, float scale, int round_out = 0, int tr_T = 0,
                                                              float* __restrict__ out2 = nullptr) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cl, g = blockIdx.y;
    ws += (long)blockIdx.z * groups * nsplit * N;
    if (blockIdx.z == 1 && out2) out = out2;  // (the second result plane has its own destination)
    else out += (long)blockIdx.z * groups * N;
    float s0 = 0.f, s1 = 0.f, c0 = 0.f, c1 = 0.f;
    if (n < N) {
        const float* w = ws + (long)g * nsplit * N + n;
        int p = pl;
        for (; p + 8 < nsplit; p += 16) {
            kahan1(s0, c0, w[(long)p * N]);
            kahan1(s1, c1, w[(long)(p + 8) * N]);
        }
        if (p < nsplit) kahan1(s0, c0, w[(long)p * N]);
    }
    kahan1(s0, c0, s1 - c1);
    red[threadIdx.x] = s0 - c0;
    __syncthreads();
    if (pl == 0 && n < N) {
        float s = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) kahan1(s, c, red[k * 32 + cl]);
        const float r = (s - c) * scale;
        const int o = tr_T > 0 ? (n % (N / tr_T)) * tr_T + n / (N / tr_T) : n;
        out[(long)g * N + o] = round_out ? rnd16(r) : r;
    }
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
    // column lanes per block: the power of two (>= 8 float4 = 128 B of a row) that leaves the fewest lanes idle in the last
    // column block - 96 float4 columns are 3 x 32, not 2 x 64 with a quarter of every block masked off
    for (int c = cl >> 1; c >= 8; c >>= 1)
        if (((nv + c - 1) / c) * c < ((nv + cl - 1) / cl) * cl) cl = c;
    const int RL = 256 / cl;
    const int cb = (nv + cl - 1) / cl;
    long want = 1024 / ((long)cb * groups);  // ~4 blocks per CU in total (more, up to 32 per CU, measured the same)
    if (want < 1) want = 1;
    long maxs = rows_per_group / ((long)RL * 8);  // >= 8 rows per thread
    if (maxs < 1) maxs = 1;
    long ns = want < maxs ? want : maxs;
    if (ns > 512) ns = 512;  // (the fold pass spreads the partials of a column over 8 lanes)
    long rps = (rows_per_group + ns - 1) / ns;
    ns = (rows_per_group + rps - 1) / rps;
    *CL = cl; *nsplit = (int)ns; *rows_per_split = (int)rps; *colblocks = cb;
}

// ---- eager attention under the 16-bit policy: the softmax between the two bf16 matmuls (include/dkd_hip.h) ---------------------
// one wave per row of N logits (N % 4 == 0, N <= 64 * 4 * VPL), everything between the bf16 load and the bf16 store in fp32
template <int VPL, bool BWD>
__global__ __launch_bounds__(256) void attn16_softmax_kernel(const __bf16* __restrict__ S, const __bf16* __restrict__ dP,
                                                             __bf16* __restrict__ out, long rows, int N, long ld, float c1, float c2,
                                                             int valid) {
    // valid < N: the key columns valid..N-1 are padding (a token count that is not a multiple of 8, ops.attention16): they are
    // masked out of the softmax and their P / dS entries are written as zeros
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nv = N >> 2;
    for (long row = (long)blockIdx.x * 4 + wid; row < rows; row += (long)gridDim.x * 4) {
        const long r4 = row * (ld >> 2);
        float4 t[VPL];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                t[i] = rnd16(ld4(S, r4 + c));  // (bf16 already; rnd16 is the identity here)
                t[i] = rnd16(make_float4(t[i].x * c1, t[i].y * c1, t[i].z * c1, t[i].w * c1));
                t[i] = rnd16(make_float4(t[i].x * c2, t[i].y * c2, t[i].z * c2, t[i].w * c2));
                if (4 * c + 3 >= valid) {
                    if (4 * c + 0 >= valid) t[i].x = -INFINITY;
                    if (4 * c + 1 >= valid) t[i].y = -INFINITY;
                    if (4 * c + 2 >= valid) t[i].z = -INFINITY;
                    t[i].w = -INFINITY;
                }
                mx = fmaxf(mx, fmaxf(fmaxf(t[i].x, t[i].y), fmaxf(t[i].z, t[i].w)));
            } else {
                t[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            t[i] = make_float4(__expf(t[i].x - mx), __expf(t[i].y - mx), __expf(t[i].z - mx), __expf(t[i].w - mx));
            sum += (t[i].x + t[i].y) + (t[i].z + t[i].w);
        }
        const float inv = 1.f / wave_sum(sum);
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) st4(out, r4 + c, make_float4(t[i].x * inv, t[i].y * inv, t[i].z * inv, t[i].w * inv));
            }
            continue;
        }
        float4 d[VPL];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            t[i] = make_float4(t[i].x * inv, t[i].y * inv, t[i].z * inv, t[i].w * inv);  // P32
            d[i] = c < nv ? ld4(dP, r4 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            dot += (d[i].x * t[i].x + d[i].y * t[i].y) + (d[i].z * t[i].z + d[i].w * t[i].w);
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 v = rnd16(make_float4(t[i].x * (d[i].x - dot), t[i].y * (d[i].y - dot), t[i].z * (d[i].z - dot), t[i].w * (d[i].w - dot)));
                v = rnd16(make_float4(v.x * c2, v.y * c2, v.z * c2, v.w * c2));
                v = rnd16(make_float4(v.x * c1, v.y * c1, v.z * c1, v.w * c1));
                st4(out, r4 + c, v);
            }
        }
    }
}

// rows regrouped: dst group b holds rows 0..gs_dst-1, row i < gs_src taken from src group b, the others zero.  16-byte vectors.
__global__ void regroup_rows16_kernel(const uint4* __restrict__ src, long lds4, int gs_src, uint4* __restrict__ dst, long ldd4,
                                      int gs_dst, long groups, int cv) {
    const long total = groups * gs_dst * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv);
        const long r = idx / cv, b = r / gs_dst;
        const int i = (int)(r % gs_dst);
        dst[r * ldd4 + c] = i < gs_src ? src[(b * gs_src + i) * lds4 + c] : make_uint4(0u, 0u, 0u, 0u);
    }
}

__global__ void scale_bf16_kernel(const __bf16* __restrict__ x, float s, __bf16* __restrict__ out, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const float4 v = ld4(x, i);
        st4(out, i, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
    }
}

}  // namespace

// (LPR, VPL) for a row of nv float4: the narrowest lane group that holds it
#define DKD_LN_DISPATCH(nv, CALL)                          \
    do {                                                   \
        if ((nv) <= 16) { CALL(16, 1); }                   \
        else if ((nv) <= 32) { CALL(32, 1); }              \
        else if ((nv) <= 64) { CALL(64, 1); }              \
        else if ((nv) <= 128) { CALL(64, 2); }             \
        else if ((nv) <= 192) { CALL(64, 3); }             \
        else if ((nv) <= 256) { CALL(64, 4); }             \
        else if ((nv) <= 384) { CALL(64, 6); }             \
        else { CALL(64, 8); }                              \
    } while (0)

static int ln_rows_per_block(int nv) { return 4 * (nv <= 16 ? 4 : (nv <= 32 ? 2 : 1)); }

extern "C" int dkd_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                     float* rstd, long rows, int C, float eps, int in_row_grp, int in_row_grp_stride,
                                     int in_row_off, void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int nv = C / 4, rpb = ln_rows_per_block(nv);
    const int blocks = (int)((rows + rpb - 1) / rpb < 4096 ? (rows + rpb - 1) / rpb : 4096);
    hipStream_t st = (hipStream_t)stream;
#define DKD_LN_FWD(L, V)                                                                                                   \
    if (mean && rstd)                                                                                                      \
        hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y, mean,    \
                           rstd, rows, C, eps, in_row_grp, in_row_grp_stride, in_row_off);                                 \
    else                                                                                                                   \
        hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y, mean,   \
                           rstd, rows, C, eps, in_row_grp, in_row_grp_stride, in_row_off)
    DKD_LN_DISPATCH(nv, DKD_LN_FWD);
#undef DKD_LN_FWD
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_layernorm_fwd_bf16out_f32(const float* x, const float* gamma, const float* beta, void* y_bf16, long rows, int C,
                                             float eps, int in_row_grp, int in_row_grp_stride, int in_row_off, void* stream) {
    if (!x || !gamma || !beta || !y_bf16 || rows <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int nv = C / 4, rpb = ln_rows_per_block(nv);
    const int blocks = (int)((rows + rpb - 1) / rpb < 4096 ? (rows + rpb - 1) / rpb : 4096);
#define DKD_LN_FWD16(L, V)                                                                                                     \
    hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma,   \
                       beta, reinterpret_cast<float*>(y_bf16), nullptr, nullptr, rows, C, eps, in_row_grp, in_row_grp_stride,  \
                       in_row_off)
    DKD_LN_DISPATCH(nv, DKD_LN_FWD16);
#undef DKD_LN_FWD16
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

// mixed-precision step: x fp32 or bf16 (x16), y fp32 or bf16 (y16); the arithmetic and the statistics are fp32 either way
// (torch autocast runs layer_norm in fp32; a bf16 y is the rounding its only consumer, a 16-bit Linear / conv, applies)
extern "C" int dkd_layernorm_fwd_io(const void* x, int x16, const float* gamma, const float* beta, void* y, int y16, float* mean,
                                    float* rstd, long rows, int C, float eps, int cf16, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    if (cf16 && !x16) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int nv = C / 4, rpb = ln_rows_per_block(nv);
    const int blocks = (int)((rows + rpb - 1) / rpb < 4096 ? (rows + rpb - 1) / rpb : 4096);
    hipStream_t st = (hipStream_t)stream;
#define DKD_LN_FWD_CF(L, V)                                                                                                  \
    if (y16) hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, true, true, __bf16, true>), dim3(blocks), dim3(256), 0, st,      \
                                reinterpret_cast<const __bf16*>(x), gamma, beta, reinterpret_cast<float*>(y), mean, rstd,    \
                                rows, C, eps, 0, 0, 0);                                                                      \
    else hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, true, false, __bf16, true>), dim3(blocks), dim3(256), 0, st,         \
                            reinterpret_cast<const __bf16*>(x), gamma, beta, reinterpret_cast<float*>(y), mean, rstd, rows,  \
                            C, eps, 0, 0, 0)
    if (cf16) {
        DKD_LN_DISPATCH(nv, DKD_LN_FWD_CF);
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
#undef DKD_LN_FWD_CF
#define DKD_LN_FWD_IO(L, V)                                                                                                  \
    DKD_IO_TYPE(x16, TI,                                                                                                     \
                if (y16) hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, true, true, TI>), dim3(blocks), dim3(256), 0, st,   \
                                            reinterpret_cast<const TI*>(x), gamma, beta, reinterpret_cast<float*>(y), mean,  \
                                            rstd, rows, C, eps, 0, 0, 0);                                                    \
                else hipLaunchKernelGGL((layernorm_fwd_kernel<L, V, true, false, TI>), dim3(blocks), dim3(256), 0, st,      \
                                        reinterpret_cast<const TI*>(x), gamma, beta, reinterpret_cast<float*>(y), mean,      \
                                        rstd, rows, C, eps, 0, 0, 0))
    DKD_LN_DISPATCH(nv, DKD_LN_FWD_IO);
#undef DKD_LN_FWD_IO
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

// 4 workgroups per CU: enough waves in flight to cover the HBM latency of the two row reads; partials stay tiny
static int ln_bwd_blocks(long rows, int C) {
    const int rpb = ln_rows_per_block(C / 4);
    const long b = (rows + rpb - 1) / rpb;
    return (int)(b < 1024 ? b : 1024);
}

extern "C" long dkd_layernorm_bwd_f32_workspace_bytes(long rows, int C) {
    return (long)ln_bwd_blocks(rows, C) * 2 * C * sizeof(float);
}

extern "C" int dkd_layernorm_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, const float* add, float* dx, float* dgamma, float* dbeta,
                                     float* ws, long rows, int C, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || rows <= 0) return DKD_ERR_BAD_ARG;
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    const int blocks = ln_bwd_blocks(rows, C), nv = C / 4;
    hipStream_t st = (hipStream_t)stream;
    if (dgamma && dbeta) {
        if (!ws) return DKD_ERR_BAD_ARG;
#define DKD_LN_BWD(L, V) \
    hipLaunchKernelGGL((layernorm_bwd_kernel<L, V, true>), dim3(blocks), dim3(256), 0, st, dy, x, mean, rstd, gamma, add, dx, ws, rows, C)
        DKD_LN_DISPATCH(nv, DKD_LN_BWD);
#undef DKD_LN_BWD
        DKD_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C / 4 + 15) / 16), dim3(256), 0, st, ws, blocks, 2 * C, dgamma,
                           dbeta, C, 1.0f);
    } else {
#define DKD_LN_BWD(L, V) \
    hipLaunchKernelGGL((layernorm_bwd_kernel<L, V, false>), dim3(blocks), dim3(256), 0, st, dy, x, mean, rstd, gamma, add, dx, ws, rows, C)
        DKD_LN_DISPATCH(nv, DKD_LN_BWD);
#undef DKD_LN_BWD
    }
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

// dy / x / dx each fp32 or bf16; dgamma / dbeta (both or neither) fp32; ws as dkd_layernorm_bwd_f32
extern "C" int dkd_layernorm_bwd_io(const void* dy, int dy16, const void* x, int x16, const float* mean, const float* rstd,
                                    const float* gamma, void* dx, int dx16, float* dgamma, float* dbeta, float* ws, long rows,
                                    int C, int cf16, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || rows <= 0) return DKD_ERR_BAD_ARG;
    if (cf16 && !(x16 && dx16 && dgamma)) return DKD_ERR_BAD_ARG;  // (bf16 x and dx, with parameter gradients)
    if ((C & 3) || C > LN_MAXV * 256) return DKD_ERR_UNSUPPORTED;
    if ((dgamma == nullptr) != (dbeta == nullptr) || (dgamma && !ws)) return DKD_ERR_BAD_ARG;
    const int blocks = ln_bwd_blocks(rows, C), nv = C / 4;
    hipStream_t st = (hipStream_t)stream;
    const bool affine = dgamma != nullptr;
#define DKD_LN_BWD_CF(L, V)                                                                                                    \
    DKD_IO_TYPE(dy16, TD,                                                                                                      \
        hipLaunchKernelGGL((layernorm_bwd_kernel<L, V, true, TD, __bf16, __bf16, true>), dim3(blocks), dim3(256), 0, st,       \
                           reinterpret_cast<const TD*>(dy), reinterpret_cast<const __bf16*>(x), mean, rstd, gamma,             \
                           (const float*)nullptr, reinterpret_cast<__bf16*>(dx), ws, rows, C))
    if (cf16) {
        DKD_LN_DISPATCH(nv, DKD_LN_BWD_CF);
        DKD_LAUNCH_CHECK();
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C / 4 + 15) / 16), dim3(256), 0, st, ws, blocks, 2 * C, dgamma, dbeta, C,
                           1.0f);
        DKD_LAUNCH_CHECK();
        return DKD_OK;
    }
#undef DKD_LN_BWD_CF
#define DKD_LN_BWD_IO(L, V)                                                                                                    \
    DKD_IO_TYPE(dy16, TD, DKD_IO_TYPE(x16, TI, DKD_IO_TYPE(dx16, TO,                                                           \
        if (affine) hipLaunchKernelGGL((layernorm_bwd_kernel<L, V, true, TD, TI, TO>), dim3(blocks), dim3(256), 0, st,         \
                                       reinterpret_cast<const TD*>(dy), reinterpret_cast<const TI*>(x), mean, rstd, gamma,     \
                                       (const float*)nullptr, reinterpret_cast<TO*>(dx), ws, rows, C);                         \
        else hipLaunchKernelGGL((layernorm_bwd_kernel<L, V, false, TD, TI, TO>), dim3(blocks), dim3(256), 0, st,               \
                                reinterpret_cast<const TD*>(dy), reinterpret_cast<const TI*>(x), mean, rstd, gamma,            \
                                (const float*)nullptr, reinterpret_cast<TO*>(dx), ws, rows, C))))
    DKD_LN_DISPATCH(nv, DKD_LN_BWD_IO);
#undef DKD_LN_BWD_IO
    DKD_LAUNCH_CHECK();
    if (affine) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C / 4 + 15) / 16), dim3(256), 0, st, ws, blocks, 2 * C, dgamma, dbeta, C,
                           1.0f);
        DKD_LAUNCH_CHECK();
    }
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
    const bool two = op == DKD_RED_DOT_SUM || op == DKD_RED_DOTGELU_SUM;
    if ((op == DKD_RED_DOT || two) && !Y) return DKD_ERR_BAD_ARG;
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
        case DKD_RED_GELUSQ: DKD_CR(DKD_RED_GELUSQ); break;
        case DKD_RED_DOTGELU_SUM: DKD_CR(DKD_RED_DOTGELU_SUM); break;
        default: return DKD_ERR_BAD_ARG;
    }
#undef DKD_CR
    DKD_LAUNCH_CHECK();
    if (ns > 1) {
        const dim3 fgrid((N + 31) / 32, groups, two ? 2 : 1);
        hipLaunchKernelGGL(colreduce_final_kernel, fgrid, dim3(256), 0, st, ws, out, groups, ns, N, scale);
        DKD_LAUNCH_CHECK();
    }
    return DKD_OK;
}

// BatchNorm backward, pass 1 of 2 (pointwise.hip holds pass 2): d gamma = sum dbn * xhat and d beta = sum dbn over the rows, straight
// from (dt, z) - no dbn / xhat tensors in HBM.  ws: dkd_colreduce_f32_workspace_bytes(rows, 1, N).
extern "C" int dkd_bn_bwd_stats_f32(const float* dt, const float* z, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, float* dgamma, float* dbeta, float* ws, long rows, int N, int relu, void* stream) {
    if (!dt || !z || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || rows <= 0 || N <= 0 || relu < 0 || (relu & ~16) > 4)
        return DKD_ERR_BAD_ARG;
    if (N & 3) return DKD_ERR_ALIGN;
    int cl, ns, rps, cb;
    colreduce_plan(rows, 1, N, &cl, &ns, &rps, &cb);
    if (ns > 1 && !ws) return DKD_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    dkd_bn_cols bn{mean, rstd, gamma, beta, relu, dbeta};
    hipLaunchKernelGGL(colreduce_kernel<DKD_RED_BN_BWD>, dim3(cb, ns, 1), dim3(256), 0, st, dt, z, (const float*)nullptr,
                       ns > 1 ? ws : dgamma, rows, N, (long)N, cl, rps, 1.0f, ns, 0, bn);
    DKD_LAUNCH_CHECK();
    if (ns > 1) {
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((N + 31) / 32, 1, 2), dim3(256), 0, st, ws, dgamma, 1, ns, N, 1.0f, 0, 0, dbeta);
        DKD_LAUNCH_CHECK();
    }
    return DKD_OK;
}

// ---- BatchNorm on a bf16 conv output (the mixed-precision ResNet keeps its activations as bf16 in HBM) ------------------------
// train-mode statistics: mean, then the centred (biased) variance - two passes, fp32 accumulation; ws as dkd_colreduce_f32
extern "C" int dkd_bn_stats_bf16(const void* z, float* mean, float* var, float* ws, long rows, int N, void* stream) {
    if (!z || !mean || !var || rows <= 0 || N <= 0) return DKD_ERR_BAD_ARG;
    if (N & 3) return DKD_ERR_ALIGN;
    int cl, ns, rps, cb;
    colreduce_plan(rows, 1, N, &cl, &ns, &rps, &cb);
    if (ns > 1 && !ws) return DKD_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const __bf16* x = reinterpret_cast<const __bf16*>(z);
    const float inv = 1.0f / (float)rows;
    for (int pass = 0; pass < 2; ++pass) {
        float* out = pass ? var : mean;
        if (pass == 0)
            hipLaunchKernelGGL((colreduce_kernel<DKD_RED_SUM, __bf16, false>), dim3(cb, ns, 1), dim3(256), 0, st, x, (const __bf16*)nullptr,
                               (const float*)nullptr, ns > 1 ? ws : out, rows, N, (long)N, cl, rps, inv, ns, 0);
        else
            hipLaunchKernelGGL((colreduce_kernel<DKD_RED_SQDIFF, __bf16, false>), dim3(cb, ns, 1), dim3(256), 0, st, x, (const __bf16*)nullptr,
                               (const float*)mean, ns > 1 ? ws : out, rows, N, (long)N, cl, rps, inv, ns, 0);
        DKD_LAUNCH_CHECK();
        if (ns > 1) {
            hipLaunchKernelGGL(colreduce_final_kernel, dim3((N + 31) / 32, 1, 1), dim3(256), 0, st, ws, out, 1, ns, N, inv);
            DKD_LAUNCH_CHECK();
        }
    }
    return DKD_OK;
}
// dkd_bn_bwd_stats_f32 on bf16 dt and z
extern "C" int dkd_bn_bwd_stats_bf16(const void* dt, const void* z, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, float* dgamma, float* dbeta, float* ws, long rows, int N, int relu, void* stream) {
    if (!dt || !z || !mean || !rstd || !gamma || !beta || !dgamma || !dbeta || rows <= 0 || N <= 0 || relu < 0 || (relu & ~16) > 4)
        return DKD_ERR_BAD_ARG;
    if (N & 3) return DKD_ERR_ALIGN;
    int cl, ns, rps, cb;
    colreduce_plan(rows, 1, N, &cl, &ns, &rps, &cb);
    if (ns > 1 && !ws) return DKD_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    dkd_bn_cols bn{mean, rstd, gamma, beta, relu, dbeta};
    hipLaunchKernelGGL((colreduce_kernel<DKD_RED_BN_BWD, __bf16, false>), dim3(cb, ns, 1), dim3(256), 0, st,
                       reinterpret_cast<const __bf16*>(dt), reinterpret_cast<const __bf16*>(z), (const float*)nullptr,
                       ns > 1 ? ws : dgamma, rows, N, (long)N, cl, rps, 1.0f, ns, 0, bn);
    DKD_LAUNCH_CHECK();
    if (ns > 1) {
        hipLaunchKernelGGL(colreduce_final_kernel, dim3((N + 31) / 32, 1, 2), dim3(256), 0, st, ws, dgamma, 1, ns, N, 1.0f, 0, 0, dbeta);
        DKD_LAUNCH_CHECK();
    }
    return DKD_OK;
}

extern "C" long dkd_colreduce_tr_f32_workspace_bytes(long rows, int N) {
    int cl, ns, rps, cb;
    colreduce_plan(rows, 1, N, &cl, &ns, &rps, &cb);
    return (long)ns * N * sizeof(float);  // (always through the fold pass, which does the transposition)
}

extern "C" int dkd_colreduce_tr_f32(const float* X, float* out, float* ws, long rows, int N, int T, int round_out, void* stream) {
    if (!X || !out || !ws || rows <= 0 || N <= 0 || T <= 0 || (N % T)) return DKD_ERR_BAD_ARG;
    if (N & 3) return DKD_ERR_ALIGN;
    int cl, ns, rps, cb;
    colreduce_plan(rows, 1, N, &cl, &ns, &rps, &cb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colreduce_kernel<DKD_RED_SUM>, dim3(cb, ns, 1), dim3(256), 0, st, X, (const float*)nullptr, (const float*)nullptr, ws,
                       rows, N, (long)N, cl, rps, 1.0f, ns, 0);  // partial sums [ns][N] (unscaled, unrounded even when ns == 1)
    DKD_LAUNCH_CHECK();
    hipLaunchKernelGGL(colreduce_final_kernel, dim3((N + 31) / 32, 1, 1), dim3(256), 0, st, ws, out, 1, ns, N, 1.0f, round_out, T);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

// bf16 X (and Y): the column sums of the mixed-precision step (bias gradients; GRN energies and GRN-backward sums with the
// GELU rounded to bf16 when round_gelu != 0).  Ops: DKD_RED_SUM, DKD_RED_DOT, DKD_RED_GELUSQ, DKD_RED_DOTGELU_SUM; fp32 results
extern "C" int dkd_colreduce_bf16(const void* X, const void* Y, float* out, float* ws, long rows_per_group, int groups, int N,
                                  int op, int round_gelu, int round_out, float scale, void* stream) {
    if (!X || !out || rows_per_group <= 0 || groups <= 0 || N <= 0) return DKD_ERR_BAD_ARG;
    if (N & 3) return DKD_ERR_ALIGN;
    const bool two = op == DKD_RED_DOTGELU_SUM;
    if ((op == DKD_RED_DOT || two) && !Y) return DKD_ERR_BAD_ARG;
    int cl, ns, rps, cb;
    colreduce_plan(rows_per_group, groups, N, &cl, &ns, &rps, &cb);
    if (ns > 1 && !ws) return DKD_ERR_BAD_ARG;
    float* dst = ns > 1 ? ws : out;
    dim3 grid(cb, ns, groups), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* x = reinterpret_cast<const __bf16*>(X);
    const __bf16* y = reinterpret_cast<const __bf16*>(Y);
#define DKD_CR16(OPV, R)                                                                                                     \
    hipLaunchKernelGGL((colreduce_kernel<OPV, __bf16, R>), grid, block, 0, st, x, y, (const float*)nullptr, dst, rows_per_group, N, \
                       (long)N, cl, rps, scale, ns, round_out)
    switch (op) {
        case DKD_RED_SUM: DKD_CR16(DKD_RED_SUM, false); break;
        case DKD_RED_DOT: DKD_CR16(DKD_RED_DOT, false); break;
        case DKD_RED_GELUSQ: if (round_gelu) DKD_CR16(DKD_RED_GELUSQ, true); else DKD_CR16(DKD_RED_GELUSQ, false); break;
        case DKD_RED_DOTGELU_SUM: if (round_gelu) DKD_CR16(DKD_RED_DOTGELU_SUM, true); else DKD_CR16(DKD_RED_DOTGELU_SUM, false); break;
        default: return DKD_ERR_UNSUPPORTED;
    }
#undef DKD_CR16
    DKD_LAUNCH_CHECK();
    if (ns > 1) {
        const dim3 fgrid((N + 31) / 32, groups, two ? 2 : 1);
        hipLaunchKernelGGL(colreduce_final_kernel, fgrid, dim3(256), 0, st, ws, out, groups, ns, N, scale, round_out);
        DKD_LAUNCH_CHECK();
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

template <bool BWD>
static int launch_attn16_softmax(const void* S, const void* dP, void* out, long rows, int N, long ld, float c1, float c2, int valid,
                                 void* stream) {
    if (!S || !out || (BWD && !dP) || rows <= 0 || N <= 0 || valid < 0 || valid > N) return DKD_ERR_BAD_ARG;
    if (valid == 0) valid = N;
    if ((N & 3) || (ld & 3) || ld < N) return DKD_ERR_ALIGN;
    if (N > 64 * 4 * 8) return DKD_ERR_UNSUPPORTED;
    const int nv = N / 4, vpl = (nv + 63) / 64;
    const int blocks = (int)((rows + 3) / 4 < 16384 ? (rows + 3) / 4 : 16384);
    const __bf16* s = reinterpret_cast<const __bf16*>(S);
    const __bf16* d = reinterpret_cast<const __bf16*>(dP);
    __bf16* o = reinterpret_cast<__bf16*>(out);
#define DKD_SM16(V) hipLaunchKernelGGL((attn16_softmax_kernel<V, BWD>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, d, o, rows, N, ld, c1, c2, valid)
    if (vpl <= 1) DKD_SM16(1); else if (vpl <= 2) DKD_SM16(2); else if (vpl <= 4) DKD_SM16(4); else if (vpl <= 6) DKD_SM16(6); else DKD_SM16(8);
#undef DKD_SM16
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_attn16_softmax_fwd(const void* S, void* P, long rows, int N, long ld, float c1, float c2, int valid, void* stream) {
    return launch_attn16_softmax<false>(S, nullptr, P, rows, N, ld, c1, c2, valid, stream);
}
extern "C" int dkd_attn16_softmax_bwd(const void* S, const void* dP, void* dS, long rows, int N, long ld, float c1, float c2, int valid,
                                      void* stream) {
    return launch_attn16_softmax<true>(S, dP, dS, rows, N, ld, c1, c2, valid, stream);
}
extern "C" int dkd_regroup_rows_bf16(const void* src, long lds, int gs_src, void* dst, long ldd, int gs_dst, long groups, int cols,
                                     void* stream) {
    if (!src || !dst || gs_src <= 0 || gs_dst <= 0 || groups <= 0 || cols <= 0 || lds < cols || ldd < cols) return DKD_ERR_BAD_ARG;
    if ((cols & 7) || (lds & 7) || (ldd & 7) || !dkd_aligned16(src) || !dkd_aligned16(dst)) return DKD_ERR_ALIGN;
    const long total = groups * gs_dst * (cols / 8);
    hipLaunchKernelGGL(regroup_rows16_kernel, dim3((int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const uint4*>(src), lds / 8, gs_src, reinterpret_cast<uint4*>(dst), ldd / 8,
                       gs_dst, groups, cols / 8);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_scale_bf16(const void* x, float s, void* out, long n, void* stream) {
    if (!x || !out || n <= 0) return DKD_ERR_BAD_ARG;
    if (n & 3) return DKD_ERR_ALIGN;
    const long nvec = n / 4;
    hipLaunchKernelGGL(scale_bf16_kernel, dim3((int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const __bf16*>(x), s, reinterpret_cast<__bf16*>(out), nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
