// Element-wise HBM-bound kernels (16 B per lane, grid-stride): activations and their derivatives, train-mode
// BatchNorm apply / backward for the ScaleKD projector input, GRN, small helpers.
#include "common.h"
#include "dkd_hip.h"

namespace {

inline int ew_blocks(long nvec) {
    long b = (nvec + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

#define EW_LOOP(i, nvec) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (nvec); i += (long)gridDim.x * blockDim.x)

__global__ void gelu_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 v = x[i];
        y[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    }
}
__global__ void gelu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ x, float4* __restrict__ dx,
                                long nvec) {
    EW_LOOP(i, nvec) {
        const float4 v = x[i], g = dy[i];
        dx[i] = make_float4(g.x * gelu_grad_f(v.x), g.y * gelu_grad_f(v.y), g.z * gelu_grad_f(v.z), g.w * gelu_grad_f(v.w));
    }
}
__global__ void relu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ dx,
                                long nvec) {
    EW_LOOP(i, nvec) {
        const float4 v = y[i], g = dy[i];
        dx[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    }
}
__global__ void axpby_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out,
                             float alpha, float beta, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 u = a[i], v = b[i];
        out[i] = make_float4(alpha * u.x + beta * v.x, alpha * u.y + beta * v.y, alpha * u.z + beta * v.z,
                             alpha * u.w + beta * v.w);
    }
}
// out[r][:] = x[r][:] - vec[r / rows_per_group][:]
__global__ void sub_group_vec_kernel(const float4* __restrict__ x, const float4* __restrict__ vec,
                                     float4* __restrict__ out, long rows_per_group, int nv, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / nv;
        const int c = (int)(i % nv);
        const float4 a = x[i], m = vec[(r / rows_per_group) * nv + c];
        out[i] = make_float4(a.x - m.x, a.y - m.y, a.z - m.z, a.w - m.w);
    }
}

// out[r][c] = x[r][c] * mul[g][c] + add[g][c], g = r / rows_per_group (mul / add nullable): squeeze-excitation's channel
// re-scaling per image and its adjoint (MobileNetV3 SElayer: scale * input; dx = dy * scale + d(pool) / HW)
__global__ void group_affine_kernel(const float4* __restrict__ x, const float4* __restrict__ mul, const float4* __restrict__ add,
                                    float4* __restrict__ out, long rows_per_group, int nv, float add_scale, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / nv;
        const int c = (int)(i % nv);
        const long gi = (r / rows_per_group) * nv + c;
        float4 v = x[i];
        if (mul) { const float4 m = mul[gi]; v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w; }
        if (add) { const float4 a = add[gi]; v.x = fmaf(a.x, add_scale, v.x); v.y = fmaf(a.y, add_scale, v.y); v.z = fmaf(a.z, add_scale, v.z); v.w = fmaf(a.w, add_scale, v.w); }
        out[i] = v;
    }
}

// nn.Hardsigmoid: relu6(x + 3) / 6; with dy: dy * (|x| < 3 ? 1/6 : 0)
__global__ void hardsigmoid_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = dy ? (v > -3.f && v < 3.f ? dy[i] * (1.f / 6.f) : 0.f) : fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
}

// SwiGLU (DINOv2 ViT-g FFN): x12 [rows, 2*Hd] -> out[rows, Hd] = silu(x1) * x2
__global__ void swiglu_fwd_kernel(const float4* __restrict__ x12, float4* __restrict__ out, int hv, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / hv;
        const int c = (int)(i % hv);
        const float4 a = x12[r * 2 * hv + c], b = x12[r * 2 * hv + hv + c];
        float4 o;
        o.x = a.x / (1.f + __expf(-a.x)) * b.x; o.y = a.y / (1.f + __expf(-a.y)) * b.y;
        o.z = a.z / (1.f + __expf(-a.z)) * b.z; o.w = a.w / (1.f + __expf(-a.w)) * b.w;
        out[i] = o;
    }
}
__device__ __forceinline__ void swiglu_grad(float a, float b, float g, float& da, float& db) {
    const float sg = 1.f / (1.f + __expf(-a));
    const float silu = a * sg;
    da = g * b * (sg + silu * (1.f - sg));
    db = g * silu;
}
__global__ void swiglu_bwd_kernel(const float4* __restrict__ dout, const float4* __restrict__ x12, float4* __restrict__ dx12,
                                  int hv, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / hv;
        const int c = (int)(i % hv);
        const float4 a = x12[r * 2 * hv + c], b = x12[r * 2 * hv + hv + c], g = dout[i];
        float4 da, db;
        swiglu_grad(a.x, b.x, g.x, da.x, db.x); swiglu_grad(a.y, b.y, g.y, da.y, db.y);
        swiglu_grad(a.z, b.z, g.z, da.z, db.z); swiglu_grad(a.w, b.w, g.w, da.w, db.w);
        dx12[r * 2 * hv + c] = da;
        dx12[r * 2 * hv + hv + c] = db;
    }
}
// window partition of a P_h x P_w token grid (losses/scalekd.py:327-335): out[b][(wy,wx,ty,tx)] = in[b][(wy*th+ty, wx*tw+tx)]
__global__ void window_permute_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int Ph, int Pw, int wh,
                                      int ww, int dv, int inverse, long nvec) {
    const int th = Ph / wh, tw = Pw / ww;
    EW_LOOP(i, nvec) {
        long r = i / dv;
        const int c = (int)(i % dv);
        const int t = (int)(r % (Ph * Pw));
        const long b = r / (Ph * Pw);
        // t = window-major index -> grid index
        const int tx = t % tw, ty = (t / tw) % th, wx = (t / (tw * th)) % ww, wy = t / (tw * th * ww);
        const long grid = (long)(wy * th + ty) * Pw + wx * tw + tx;
        const long a = (b * Ph * Pw + t) * dv + c, g = (b * Ph * Pw + grid) * dv + c;
        if (inverse) out[g] = in[a];
        else out[a] = in[g];
    }
}


// ---- BatchNorm (train mode) ------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ var, float* __restrict__ rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int N, float count,
                                   float momentum, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    rstd[n] = rsqrtf(var[n] + eps);
    if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean[n];
    if (running_var) running_var[n] = (1.f - momentum) * running_var[n] + momentum * var[n] * (count / (count - 1.f));
}

// out = act(gamma * (z - mean) * rstd + beta) + pos[row % period]
__global__ void bn_act_pos_fwd_kernel(const float4* __restrict__ z, const float4* __restrict__ mean,
                                      const float4* __restrict__ rstd, const float4* __restrict__ gamma,
                                      const float4* __restrict__ beta, const float4* __restrict__ pos,
                                      float4* __restrict__ out, int nv, int period, int relu, long nvec) {
    // (relu & 16, the mixed-precision projector: BatchNorm and the activation produce 16-bit tensors - rounded after each - before
    // the fp32 position table is added)
    const bool r16 = (relu & 16) != 0;
    relu &= 15;
    EW_LOOP(i, nvec) {
        const long r = i / nv;
        const int c = (int)(i % nv);
        const float4 x = z[i], mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
        float4 o = make_float4(g.x * (x.x - mu.x) * rs.x + b.x, g.y * (x.y - mu.y) * rs.y + b.y,
                               g.z * (x.z - mu.z) * rs.z + b.z, g.w * (x.w - mu.w) * rs.w + b.w);
        if (r16) o = rnd16(o);
        if (relu == 3) {  // SiLU (DarkNet's Conv, models/backbones/darknet.py:47)
            o = make_float4(o.x * sigmoid_f(o.x), o.y * sigmoid_f(o.y), o.z * sigmoid_f(o.z), o.w * sigmoid_f(o.w));
        } else if (relu == 4) {  // Hardswish (MobileNetV3's "HS" blocks)
            o = make_float4(hswish_f(o.x), hswish_f(o.y), hswish_f(o.z), hswish_f(o.w));
        } else if (relu) {
            o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            if (relu == 2) o = make_float4(fminf(o.x, 6.f), fminf(o.y, 6.f), fminf(o.z, 6.f), fminf(o.w, 6.f));  // ReLU6
        }
        if (r16 && relu >= 3) o = rnd16(o);   // (a smooth activation of a 16-bit tensor returns a 16-bit tensor; ReLU / ReLU6 are exact)
        if (pos) {
            const float4 p = pos[(r % period) * nv + c];
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        out[i] = o;
    }
}

// dbn = dt * act'(bn) (bn recomputed; act' = [0 < bn (< 6)] for ReLU / ReLU6), xhat = (z - mean) * rstd
__global__ void bn_bwd_prep_kernel(const float4* __restrict__ dt, const float4* __restrict__ z,
                                   const float4* __restrict__ mean, const float4* __restrict__ rstd,
                                   const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                   float4* __restrict__ dbn, float4* __restrict__ xhat, int nv, int relu, long nvec) {
    EW_LOOP(i, nvec) {
        const int c = (int)(i % nv);
        const float4 x = z[i], mu = mean[c], rs = rstd[c], g = gamma[c], b = beta[c], d = dt[i];
        const float4 xh = make_float4((x.x - mu.x) * rs.x, (x.y - mu.y) * rs.y, (x.z - mu.z) * rs.z, (x.w - mu.w) * rs.w);
        float4 o = d;
        if (relu == 3) {  // SiLU: d/dy [y s(y)] = s (1 + y (1 - s))
            const float4 y = make_float4(g.x * xh.x + b.x, g.y * xh.y + b.y, g.z * xh.z + b.z, g.w * xh.w + b.w);
            o = make_float4(d.x * silu_grad_f(y.x), d.y * silu_grad_f(y.y), d.z * silu_grad_f(y.z), d.w * silu_grad_f(y.w));
        } else if (relu == 4) {
            const float4 y = make_float4(g.x * xh.x + b.x, g.y * xh.y + b.y, g.z * xh.z + b.z, g.w * xh.w + b.w);
            o = make_float4(d.x * hswish_grad_f(y.x), d.y * hswish_grad_f(y.y), d.z * hswish_grad_f(y.z), d.w * hswish_grad_f(y.w));
        } else if (relu) {  // 1: ReLU, 2: ReLU6 (gradient 1 strictly inside (0, 6), as F.relu6 / hardtanh)
            const float hi = relu == 2 ? 6.f : INFINITY;
            const float4 y = make_float4(g.x * xh.x + b.x, g.y * xh.y + b.y, g.z * xh.z + b.z, g.w * xh.w + b.w);
            o.x = y.x > 0.f && y.x < hi ? d.x : 0.f;
            o.y = y.y > 0.f && y.y < hi ? d.y : 0.f;
            o.z = y.z > 0.f && y.z < hi ? d.z : 0.f;
            o.w = y.w > 0.f && y.w < hi ? d.w : 0.f;
        }
        dbn[i] = o;
        xhat[i] = xh;
    }
}

// dz = gamma * rstd * (dbn - sum_dbn / M - xhat * sum_dbn_xhat / M)
__global__ void bn_bwd_apply_kernel(const float4* __restrict__ dbn, const float4* __restrict__ xhat,
                                    const float4* __restrict__ s1, const float4* __restrict__ s2,
                                    const float4* __restrict__ gamma, const float4* __restrict__ rstd,
                                    float4* __restrict__ dz, int nv, float inv_count, long nvec) {
    EW_LOOP(i, nvec) {
        const int c = (int)(i % nv);
        const float4 d = dbn[i], xh = xhat[i], a = s1[c], b = s2[c], g = gamma[c], rs = rstd[c];
        dz[i] = make_float4(g.x * rs.x * (d.x - a.x * inv_count - xh.x * b.x * inv_count),
                            g.y * rs.y * (d.y - a.y * inv_count - xh.y * b.y * inv_count),
                            g.z * rs.z * (d.z - a.z * inv_count - xh.z * b.z * inv_count),
                            g.w * rs.w * (d.w - a.w * inv_count - xh.w * b.w * inv_count));
    }
}

// pass 2 of the BatchNorm backward without intermediates: dbn and xhat recomputed from (dt, z), dz fp32 or bf16 (the mixed-precision
// step: BatchNorm's input is a 16-bit tensor, so its gradient is rounded)
template <typename TO, typename TI = float>
__global__ void bn_bwd_apply_direct_kernel(const TI* __restrict__ dt, const TI* __restrict__ z, dkd_bn_cols bn,
                                           const float* __restrict__ s1, const float* __restrict__ s2, TO* __restrict__ dz, int nv,
                                           float inv_count, long nvec) {
    EW_LOOP(i, nvec) {
        const int c = (int)(i % nv);
        const float4 x = ld4(z, i), mu = ld4(bn.mean, c), rs = ld4(bn.rstd, c), g = ld4(bn.gamma, c), bt = ld4(bn.beta, c);
        const float4 a = ld4(s1, c), b = ld4(s2, c);
        float4 d = ld4(dt, i);
        const float4 xh = make_float4((x.x - mu.x) * rs.x, (x.y - mu.y) * rs.y, (x.z - mu.z) * rs.z, (x.w - mu.w) * rs.w);
        if (bn.act) {
            d.x = bn_act_grad(d.x, g.x * xh.x + bt.x, bn.act); d.y = bn_act_grad(d.y, g.y * xh.y + bt.y, bn.act);
            d.z = bn_act_grad(d.z, g.z * xh.z + bt.z, bn.act); d.w = bn_act_grad(d.w, g.w * xh.w + bt.w, bn.act);
        }
        st4(dz, i, make_float4(g.x * rs.x * (d.x - a.x * inv_count - xh.x * b.x * inv_count),
                               g.y * rs.y * (d.y - a.y * inv_count - xh.y * b.y * inv_count),
                               g.z * rs.z * (d.z - a.z * inv_count - xh.z * b.z * inv_count),
                               g.w * rs.w * (d.w - a.w * inv_count - xh.w * b.w * inv_count)));
    }
}

// BatchNorm + activation on a bf16 conv output -> bf16 (the policy: BatchNorm returns a 16-bit tensor - rounded - and the activation
// acts on it); ReLU / ReLU6 / none
__global__ void bn_act_fwd_bf16_kernel(const __bf16* __restrict__ z, dkd_bn_cols bn, __bf16* __restrict__ out, int nv, long nvec) {
    EW_LOOP(i, nvec) {
        const int c = (int)(i % nv);
        const float4 x = ld4(z, i), mu = ld4(bn.mean, c), rs = ld4(bn.rstd, c), g = ld4(bn.gamma, c), b = ld4(bn.beta, c);
        float4 o = rnd16(make_float4(g.x * (x.x - mu.x) * rs.x + b.x, g.y * (x.y - mu.y) * rs.y + b.y,
                                     g.z * (x.z - mu.z) * rs.z + b.z, g.w * (x.w - mu.w) * rs.w + b.w));
        if (bn.act) {
            o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            if (bn.act == 2) o = make_float4(fminf(o.x, 6.f), fminf(o.y, 6.f), fminf(o.z, 6.f), fminf(o.w, 6.f));
        }
        st4(out, i, o);
    }
}
// relu(rnd(a + b)) on bf16 tensors: the residual add of a 16-bit block output and its 16-bit shortcut, then ReLU (resnet.py:395-397)
__global__ void add_relu_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ out, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 x = ld4(a, i), y = ld4(b, i);
        st4(out, i, make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f)));
    }
}

// ---- GRN (ConvNeXt-V2), NHWC: rows grouped per image ------------------------------------------------------
// nx[b][c] = sqrt(sumsq[b][c]) / (mean_c sqrt(sumsq[b][:]) + 1e-6); one block per image
__global__ __launch_bounds__(256) void grn_stats_kernel(const float* __restrict__ sumsq, float* __restrict__ gx,
                                                        float* __restrict__ nx, float* __restrict__ denom, int C) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = sqrtf(sumsq[(long)b * C + c]);
        gx[(long)b * C + c] = g;
        s += g;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float d = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C + 1e-6f;
    if (threadIdx.x == 0) denom[b] = d;
    for (int c = threadIdx.x; c < C; c += 256) nx[(long)b * C + c] = gx[(long)b * C + c] / d;
}

// y = g * (gamma * nx[b] + 1) + beta;  PREACT: the input is the pre-activation h and g = gelu(h) is recomputed here (the
// block then never stores g: one [M,4C] write less in the producing GEMM, one read less in the backward)
// T: element type of g / y (float, or __bf16 in the mixed-precision step: gelu(h) is then rounded to bf16 as the 16-bit GELU
// output it stands for, the GRN arithmetic is fp32 as under autocast, y is rounded by its store - the cast of the next Linear)
template <bool PREACT, typename T = float>
__global__ void grn_apply_kernel(const T* __restrict__ g, const float4* __restrict__ nx,
                                 const float4* __restrict__ gamma, const float4* __restrict__ beta,
                                 T* __restrict__ y, long rows_per_img, int nv, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / nv;
        const int c = (int)(i % nv);
        float4 v = ld4(g, i);
        if (PREACT) v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
        if (PREACT && dkd_is16<T>::value) v = rnd16(v);
        const float4 n = nx[(r / rows_per_img) * nv + c], ga = gamma[c], be = beta[c];
        st4(y, i, make_float4(v.x * (ga.x * n.x + 1.f) + be.x, v.y * (ga.y * n.y + 1.f) + be.y,
                              v.z * (ga.z * n.z + 1.f) + be.z, v.w * (ga.w * n.w + 1.f) + be.w));
    }
}

// GRN backward, per image b and channel c, given A[b][c] = sum_hw dy*g :
//   dnx = gamma * A;  dgx = dnx / denom - (1/C) * sum_c'(dnx[c'] * gx[c']) / denom^2;  coef = dgx / gx (0 if gx == 0)
// Blocks [0, B): one per image, coef[b][c] (so that dg = dy*(gamma*nx+1) + coef*g).  Blocks [B, B + ceil(C/64)): the parameter
// gradients dgamma[c] = sum_b A[b][c] nx[b][c], dbeta[c] = sum_b S[b][c] with S[b][c] = sum_hw dy, in a fixed order - what used to
// be two more column reductions (four launches) per block.
__global__ __launch_bounds__(256) void grn_bwd_stats_kernel(const float* __restrict__ A, const float* __restrict__ S,
                                                            const float* __restrict__ gx, const float* __restrict__ nx,
                                                            const float* __restrict__ denom, const float* __restrict__ gamma,
                                                            float* __restrict__ coef, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int B, int C) {
    __shared__ float red[4];
    __shared__ float pg[4][64], pb[4][64];
    if ((int)blockIdx.x >= B) {
        // 64 channels x 4 image lanes per block: lane q adds the images q, q + 4, ... (loads of four images in flight), the four
        // partial sums meet in LDS and are added in lane order - a fixed order, and 16 load rounds instead of a chain of B
        const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
        const int c = ((int)blockIdx.x - B) * 64 + cl;
        float sg = 0.f, sb = 0.f;
        if (c < C) {
            int b = q;
            for (; b + 12 < B; b += 16) {
                const long i0 = (long)b * C + c, i1 = i0 + 4L * C, i2 = i0 + 8L * C, i3 = i0 + 12L * C;
                const float a0 = A[i0], a1 = A[i1], a2 = A[i2], a3 = A[i3];
                const float n0 = nx[i0], n1 = nx[i1], n2 = nx[i2], n3 = nx[i3];
                sg += a0 * n0; sg += a1 * n1; sg += a2 * n2; sg += a3 * n3;
                if (S) { sb += S[i0]; sb += S[i1]; sb += S[i2]; sb += S[i3]; }
            }
            for (; b < B; b += 4) {
                const long i = (long)b * C + c;
                sg += A[i] * nx[i];
                if (S) sb += S[i];
            }
        }
        pg[q][cl] = sg;
        pb[q][cl] = sb;
        __syncthreads();
        if (q == 0 && c < C) {
            dgamma[c] = ((pg[0][cl] + pg[1][cl]) + pg[2][cl]) + pg[3][cl];
            if (S && dbeta) dbeta[c] = ((pb[0][cl] + pb[1][cl]) + pb[2][cl]) + pb[3][cl];
        }
        return;
    }
    const int b = blockIdx.x;
    const float d = denom[b];
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += gamma[c] * A[(long)b * C + c] * gx[(long)b * C + c];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float corr = ((red[0] + red[1]) + (red[2] + red[3])) / ((float)C * d * d);
    for (int c = threadIdx.x; c < C; c += 256) {
        const long i = (long)b * C + c;
        const float a = A[i], g = gx[i];
        const float dgx = gamma[c] * a / d - corr;
        coef[i] = g > 0.f ? dgx / g : 0.f;
    }
}

// dh = (dy * (gamma*nx + 1) + coef * g) * gelu'(h);  g == nullptr: g = gelu(h) comes out of the same erf as gelu'(h)
// T (mixed-precision step, __bf16): g = rnd(gelu(h)); the gradient of the 16-bit g is rounded before the GELU derivative
// multiplies it, and dh is rounded by its store
template <bool HAVE_G, typename T = float>
__global__ void grn_gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ g,
                                    const T* __restrict__ h, const float4* __restrict__ nx,
                                    const float4* __restrict__ coef, const float4* __restrict__ gamma,
                                    T* __restrict__ dh, long rows_per_img, int nv, long nvec) {
    EW_LOOP(i, nvec) {
        const long r = i / nv;
        const int c = (int)(i % nv);
        const long bi = (r / rows_per_img) * nv + c;
        const float4 d = ld4(dy, i), hv = ld4(h, i), n = nx[bi], cf = coef[bi], ga = gamma[c];
        float4 gv, gp;
        gelu_both_f(hv.x, gv.x, gp.x);
        gelu_both_f(hv.y, gv.y, gp.y);
        gelu_both_f(hv.z, gv.z, gp.z);
        gelu_both_f(hv.w, gv.w, gp.w);
        if (HAVE_G) gv = ld4(g, i);
        else if (dkd_is16<T>::value) gv = rnd16(gv);
        float4 dg;
        if (dkd_is16<T>::value) {
            // autograd hands the 16-bit g three 16-bit gradients - from "+ x", from "x * Nx" and through the norm - and adds them
            // as they arrive, rounding each sum: rnd(rnd(d + rnd((d gamma) Nx)) + rnd(coef g))
            const float4 b = rnd16(make_float4((d.x * ga.x) * n.x, (d.y * ga.y) * n.y, (d.z * ga.z) * n.z, (d.w * ga.w) * n.w));
            const float4 s1 = rnd16(make_float4(d.x + b.x, d.y + b.y, d.z + b.z, d.w + b.w));
            const float4 c3 = rnd16(make_float4(cf.x * gv.x, cf.y * gv.y, cf.z * gv.z, cf.w * gv.w));
            dg = rnd16(make_float4(s1.x + c3.x, s1.y + c3.y, s1.z + c3.z, s1.w + c3.w));
        } else {
            dg = make_float4(d.x * (ga.x * n.x + 1.f) + cf.x * gv.x, d.y * (ga.y * n.y + 1.f) + cf.y * gv.y,
                             d.z * (ga.z * n.z + 1.f) + cf.z * gv.z, d.w * (ga.w * n.w + 1.f) + cf.w * gv.w);
        }
        st4(dh, i, make_float4(dg.x * gp.x, dg.y * gp.y, dg.z * gp.z, dg.w * gp.w));
    }
}

// bf16 <-> fp32 plumbing of the mixed-precision step
__global__ void cast_f32_kernel(const __bf16* __restrict__ in, float* __restrict__ out, long nvec) {
    EW_LOOP(i, nvec) st4(out, i, ld4(in, i));
}
__global__ void round_bf16_kernel(float* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = rnd16(x[i]);
}
__global__ void round_bf16_copy_kernel(const float* __restrict__ x, float* __restrict__ y, long nvec) {
    EW_LOOP(i, nvec) st4(y, i, rnd16(ld4(x, i)));
}
__global__ void relu_bwd_bf16_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ y, __bf16* __restrict__ dx, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 v = ld4(y, i), g = ld4(dy, i);
        st4(dx, i, make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f));
    }
}
// ---- the frequency loss's DCT under the 16-bit policy ---------------------------------------------------------------------------
// losses/scalekd.py:337-428 implements the 2-D DCT / inverse DCT as nn.Linear layers (R x R matrices, R = the token grid: 16 at
// 224 px, 32 at 448, 37 at 518) over the last axis, a
// transpose, and the same Linear again: under autocast each is a 16-bit matmul with a 16-bit result.  One step on an NHWC map
// [n, a, b, c] (a, b = R x R spatial or frequency indices): out[.., i, ..] = rnd(sum_j Mx[i][j] in[.., j, ..]) along axis b
// (axis = 0) or a (axis = 1); Mx holds the 16-bit matrix values in fp32; transposed = use Mx^T (the adjoint, for the backward
// pass); zero_in / zero_out: the DC coefficient [a = 0, b = 0] is zero (scalekd.py:114-115 and its adjoint).
template <typename TI, typename TO, int R>
__global__ __launch_bounds__(256) void dct_step_kernel(const TI* __restrict__ in, TO* __restrict__ out, const float* __restrict__ Mx,
                                                       long lines, int C, int axis, int transposed, int zero_in, int zero_out) {
    __shared__ float m[R][R];
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) m[e / R][e % R] = transposed ? Mx[(e % R) * R + e / R] : Mx[e];
    __syncthreads();
    const int cq = C >> 2;
    const long total = lines * cq;  // a line = the R elements along the contracted axis, for one (n, other index)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cq);
        const long line = idx / cq;
        const long n = line / R;
        const int o = (int)(line % R);                       // the index that is NOT contracted
        // 4-element index of (n, a, b, c4): ((n * R + a) * R + b) * cq + c4
        const long base = axis == 0 ? ((n * R + o) * R) * cq + c4 : (n * R * R + o) * cq + c4;
        const long step = axis == 0 ? cq : (long)R * cq;
        float4 acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int j = 0; j < R; ++j) {
            float4 v = rnd16(ld4(in, base + j * step));        // (an fp32 input is the Linear's 16-bit operand after its cast)
            if (zero_in && o == 0 && j == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float w = m[i][j];
                acc[i].x = fmaf(w, v.x, acc[i].x); acc[i].y = fmaf(w, v.y, acc[i].y);
                acc[i].z = fmaf(w, v.z, acc[i].z); acc[i].w = fmaf(w, v.w, acc[i].w);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float4 r = rnd16(acc[i]);
            if (zero_out && o == 0 && i == 0) r = make_float4(0.f, 0.f, 0.f, 0.f);
            st4(out, base + i * step, r);
        }
    }
}

// F.gelu on a 16-bit tensor and its backward (fp32 inside, 16-bit results): the DINOv2 MLP under the policy
__global__ void gelu_bf16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, __bf16* __restrict__ out, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 v = ld4(x, i);
        if (dy) {
            const float4 g = ld4(dy, i);
            st4(out, i, make_float4(g.x * gelu_grad_f(v.x), g.y * gelu_grad_f(v.y), g.z * gelu_grad_f(v.z), g.w * gelu_grad_f(v.w)));
        } else {
            st4(out, i, make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w)));
        }
    }
}
// out = rnd(a + b) on bf16 tensors (autograd's accumulation of two 16-bit gradients)
__global__ void add_bf16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ out, long nvec) {
    EW_LOOP(i, nvec) {
        const float4 u = ld4(a, i), v = ld4(b, i);
        st4(out, i, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
    }
}

}  // namespace

#define EW_CHECK(n) if ((n) <= 0) return DKD_ERR_BAD_ARG; if ((n) & 3) return DKD_ERR_ALIGN;
#define F4(p) reinterpret_cast<const float4*>(p)
#define F4W(p) reinterpret_cast<float4*>(p)

extern "C" int dkd_gelu_fwd_f32(const float* x, float* y, long n, void* stream) {
    EW_CHECK(n);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, F4(x), F4W(y), n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_gelu_bwd_f32(const float* dy, const float* x, float* dx, long n, void* stream) {
    EW_CHECK(n);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, F4(dy), F4(x), F4W(dx),
                       n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_relu_bwd_f32(const float* dy, const float* y, float* dx, long n, void* stream) {
    EW_CHECK(n);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, F4(dy), F4(y), F4W(dx),
                       n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_axpby_f32(const float* a, const float* b, float* out, float alpha, float beta, long n, void* stream) {
    EW_CHECK(n);
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, F4(a), F4(b), F4W(out),
                       alpha, beta, n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_sub_group_vec_f32(const float* x, const float* vec, float* out, long rows_per_group, int groups, int N,
                                     void* stream) {
    EW_CHECK(N);
    const long nvec = rows_per_group * groups * (N / 4);
    hipLaunchKernelGGL(sub_group_vec_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(x), F4(vec),
                       F4W(out), rows_per_group, N / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_group_affine_f32(const float* x, const float* mul, const float* add, float* out, long rows_per_group, int groups,
                                    int N, float add_scale, void* stream) {
    EW_CHECK(N);
    if (!x || !out || rows_per_group <= 0 || groups <= 0) return DKD_ERR_BAD_ARG;
    const long nvec = rows_per_group * groups * (N / 4);
    hipLaunchKernelGGL(group_affine_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(x), F4(mul), F4(add),
                       F4W(out), rows_per_group, N / 4, add_scale, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_hardsigmoid_f32(const float* x, const float* dy, float* out, long n, void* stream) {
    if (!x || !out || n <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(hardsigmoid_kernel, dim3((int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
                       (hipStream_t)stream, x, dy, out, n);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_finalize_f32(const float* mean, const float* var, float* rstd, float* running_mean,
                                   float* running_var, int N, long count, float momentum, float eps, void* stream) {
    if (!mean || !var || !rstd || N <= 0 || count < 2) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, rstd,
                       running_mean, running_var, N, (float)count, momentum, eps);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_act_pos_fwd_f32(const float* z, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, const float* pos, float* out, long rows, int N, int period,
                                      int relu, void* stream) {
    EW_CHECK(N);
    if (rows <= 0 || (pos && period <= 0)) return DKD_ERR_BAD_ARG;
    const long nvec = rows * (N / 4);
    hipLaunchKernelGGL(bn_act_pos_fwd_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(z), F4(mean),
                       F4(rstd), F4(gamma), F4(beta), F4(pos), F4W(out), N / 4, period, relu, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_bwd_prep_f32(const float* dt, const float* z, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, float* dbn, float* xhat, long rows, int N,
                                   int relu, void* stream) {
    EW_CHECK(N);
    const long nvec = rows * (N / 4);
    hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(dt), F4(z),
                       F4(mean), F4(rstd), F4(gamma), F4(beta), F4W(dbn), F4W(xhat), N / 4, relu, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_bwd_apply_f32(const float* dbn, const float* xhat, const float* sum_dbn, const float* sum_dbn_xhat,
                                    const float* gamma, const float* rstd, float* dz, long rows, int N, void* stream) {
    EW_CHECK(N);
    const long nvec = rows * (N / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(dbn), F4(xhat),
                       F4(sum_dbn), F4(sum_dbn_xhat), F4(gamma), F4(rstd), F4W(dz), N / 4, 1.0f / (float)rows, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_bwd_apply_direct_f32(const float* dt, const float* z, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, const float* sum_dbn, const float* sum_dbn_xhat, void* dz, int dz16, long rows,
                                          int N, int relu, void* stream) {
    EW_CHECK(N);
    if (!dt || !z || !mean || !rstd || !gamma || !beta || !sum_dbn || !sum_dbn_xhat || !dz || rows <= 0 || relu < 0 || (relu & ~16) > 4)
        return DKD_ERR_BAD_ARG;
    const long nvec = rows * (N / 4);
    dkd_bn_cols bn{mean, rstd, gamma, beta, relu, nullptr};
    DKD_IO_TYPE(dz16, TO, hipLaunchKernelGGL((bn_bwd_apply_direct_kernel<TO>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream,
                                             dt, z, bn, sum_dbn, sum_dbn_xhat, reinterpret_cast<TO*>(dz), N / 4, 1.0f / (float)rows, nvec));
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_bwd_apply_direct_bf16(const void* dt, const void* z, const float* mean, const float* rstd, const float* gamma,
                                           const float* beta, const float* sum_dbn, const float* sum_dbn_xhat, void* dz, long rows, int N,
                                           int relu, void* stream) {
    EW_CHECK(N);
    if (!dt || !z || !mean || !rstd || !gamma || !beta || !sum_dbn || !sum_dbn_xhat || !dz || rows <= 0 || relu < 0 || (relu & ~16) > 4)
        return DKD_ERR_BAD_ARG;
    const long nvec = rows * (N / 4);
    dkd_bn_cols bn{mean, rstd, gamma, beta, relu, nullptr};
    hipLaunchKernelGGL((bn_bwd_apply_direct_kernel<__bf16, __bf16>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(dt), reinterpret_cast<const __bf16*>(z), bn, sum_dbn, sum_dbn_xhat,
                       reinterpret_cast<__bf16*>(dz), N / 4, 1.0f / (float)rows, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_bn_act_fwd_bf16(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, void* out,
                                   long rows, int N, int relu, void* stream) {
    EW_CHECK(N);
    if (!z || !mean || !rstd || !gamma || !beta || !out || rows <= 0 || relu < 0 || relu > 2) return DKD_ERR_BAD_ARG;
    const long nvec = rows * (N / 4);
    dkd_bn_cols bn{mean, rstd, gamma, beta, relu, nullptr};
    hipLaunchKernelGGL(bn_act_fwd_bf16_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(z),
                       bn, reinterpret_cast<__bf16*>(out), N / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_add_relu_bf16(const void* a, const void* b, void* out, long n, void* stream) {
    EW_CHECK(n);
    if (!a || !b || !out) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_relu_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(a),
                       reinterpret_cast<const __bf16*>(b), reinterpret_cast<__bf16*>(out), n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_grn_stats_f32(const float* sumsq, float* gx, float* nx, float* denom, int B, int C, void* stream) {
    if (!sumsq || !gx || !nx || !denom || B <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(grn_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sumsq, gx, nx, denom, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_grn_apply_f32(const float* g, const float* nx, const float* gamma, const float* beta, float* y, int B,
                                 long rows_per_img, int C, int input_is_preact, void* stream) {
    EW_CHECK(C);
    if (!g || !nx || !gamma || !beta || !y || B <= 0 || rows_per_img <= 0) return DKD_ERR_BAD_ARG;
    const long nvec = (long)B * rows_per_img * (C / 4);
    if (input_is_preact)
        hipLaunchKernelGGL((grn_apply_kernel<true, float>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, g, F4(nx),
                           F4(gamma), F4(beta), y, rows_per_img, C / 4, nvec);
    else
        hipLaunchKernelGGL((grn_apply_kernel<false, float>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, g, F4(nx),
                           F4(gamma), F4(beta), y, rows_per_img, C / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_grn_bwd_stats_f32(const float* A, const float* S, const float* gx, const float* nx, const float* denom,
                                     const float* gamma, float* coef, float* dgamma, float* dbeta, int B, int C, void* stream) {
    if (!A || !gx || !nx || !denom || !gamma || !coef || !dgamma || B <= 0 || C <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(grn_bwd_stats_kernel, dim3(B + (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, A, S, gx, nx, denom, gamma,
                       coef, dgamma, dbeta, B, C);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_grn_gelu_bwd_f32(const float* dy, const float* g, const float* h, const float* nx, const float* coef,
                                    const float* gamma, float* dh, int B, long rows_per_img, int C, void* stream) {
    EW_CHECK(C);
    if (!dy || !h || !nx || !coef || !gamma || !dh || B <= 0 || rows_per_img <= 0) return DKD_ERR_BAD_ARG;
    const long nvec = (long)B * rows_per_img * (C / 4);
    if (g)
        hipLaunchKernelGGL((grn_gelu_bwd_kernel<true, float>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, dy, g, h,
                           F4(nx), F4(coef), F4(gamma), dh, rows_per_img, C / 4, nvec);
    else
        hipLaunchKernelGGL((grn_gelu_bwd_kernel<false, float>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, dy, g, h,
                           F4(nx), F4(coef), F4(gamma), dh, rows_per_img, C / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

// ---- the same two on bf16 activations (mixed-precision step): h, y, dy, dh bf16; statistics / parameters fp32 ----------------
extern "C" int dkd_grn_apply_bf16(const void* h, const float* nx, const float* gamma, const float* beta, void* y, int B,
                                  long rows_per_img, int C, void* stream) {
    EW_CHECK(C);
    if (!h || !nx || !gamma || !beta || !y || B <= 0 || rows_per_img <= 0) return DKD_ERR_BAD_ARG;
    const long nvec = (long)B * rows_per_img * (C / 4);
    hipLaunchKernelGGL((grn_apply_kernel<true, __bf16>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(h), F4(nx), F4(gamma), F4(beta), reinterpret_cast<__bf16*>(y), rows_per_img,
                       C / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_grn_gelu_bwd_bf16(const void* dy, const void* h, const float* nx, const float* coef, const float* gamma, void* dh,
                                     int B, long rows_per_img, int C, void* stream) {
    EW_CHECK(C);
    if (!dy || !h || !nx || !coef || !gamma || !dh || B <= 0 || rows_per_img <= 0) return DKD_ERR_BAD_ARG;
    const long nvec = (long)B * rows_per_img * (C / 4);
    hipLaunchKernelGGL((grn_gelu_bwd_kernel<false, __bf16>), dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(dy), (const __bf16*)nullptr, reinterpret_cast<const __bf16*>(h), F4(nx),
                       F4(coef), F4(gamma), reinterpret_cast<__bf16*>(dh), rows_per_img, C / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_cast_f32_from_bf16(const void* in, float* out, long n, void* stream) {
    EW_CHECK(n);
    if (!in || !out) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_f32_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(in),
                       out, n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_round_bf16_f32(float* x, long n, void* stream) {
    if (!x || n <= 0) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(round_bf16_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_round_bf16_copy_f32(const float* x, float* y, long n, void* stream) {
    EW_CHECK(n);
    if (!x || !y) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(round_bf16_copy_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_dct_step(const void* in, int in16, void* out, int out16, const float* Mx, long n_maps, int R, int C, int axis,
                            int transposed, int zero_in, int zero_out, void* stream) {
    EW_CHECK(C);
    if (!in || !out || !Mx || n_maps <= 0 || (axis & ~1)) return DKD_ERR_BAD_ARG;
    if (R != 16 && R != 32 && R != 37) return DKD_ERR_UNSUPPORTED;
    const long lines = n_maps * R, total = lines * (C / 4);
    int rc = DKD_OK;
#define DKD_DCT_LAUNCH(RR)                                                                                                      \
    DKD_IO_TYPE(in16, TI, DKD_IO_TYPE(out16, TO,                                                                                \
        hipLaunchKernelGGL((dct_step_kernel<TI, TO, RR>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream,            \
                           reinterpret_cast<const TI*>(in), reinterpret_cast<TO*>(out), Mx, lines, C, axis, transposed, zero_in, \
                           zero_out)))
    if (R == 16) { DKD_DCT_LAUNCH(16); }
    else if (R == 32) { DKD_DCT_LAUNCH(32); }
    else { DKD_DCT_LAUNCH(37); }
#undef DKD_DCT_LAUNCH
    DKD_LAUNCH_CHECK();
    return rc;
}
extern "C" int dkd_dct16_step(const void* in, int in16, void* out, int out16, const float* Mx, long n_maps, int C, int axis,
                              int transposed, int zero_in, int zero_out, void* stream) {
    return dkd_dct_step(in, in16, out, out16, Mx, n_maps, 16, C, axis, transposed, zero_in, zero_out, stream);
}
extern "C" int dkd_gelu_bf16(const void* x, const void* dy, void* out, long n, void* stream) {
    EW_CHECK(n);
    if (!x || !out) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(gelu_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(dy), reinterpret_cast<__bf16*>(out), n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_relu_bwd_bf16(const void* dy, const void* y, void* dx, long n, void* stream) {
    EW_CHECK(n);
    if (!dy || !y || !dx) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(relu_bwd_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(dy),
                       reinterpret_cast<const __bf16*>(y), reinterpret_cast<__bf16*>(dx), n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_add_bf16(const void* a, const void* b, void* out, long n, void* stream) {
    EW_CHECK(n);
    if (!a || !b || !out) return DKD_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(a),
                       reinterpret_cast<const __bf16*>(b), reinterpret_cast<__bf16*>(out), n / 4);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}

extern "C" int dkd_swiglu_f32(const float* x12, const float* dout, float* out, long rows, int hidden, int backward, void* stream) {
    // forward: out[rows, hidden] = silu(x12[:, :hidden]) * x12[:, hidden:];  backward: out = dx12 [rows, 2*hidden]
    if (!x12 || !out || rows <= 0 || hidden <= 0 || (backward && !dout)) return DKD_ERR_BAD_ARG;
    if (hidden & 3) return DKD_ERR_ALIGN;
    const long nvec = rows * (hidden / 4);
    if (backward)
        hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(dout), F4(x12),
                           F4W(out), hidden / 4, nvec);
    else
        hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(x12), F4W(out),
                           hidden / 4, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
extern "C" int dkd_window_permute_f32(const float* in, float* out, int B, int Ph, int Pw, int win_h, int win_w, int D,
                                      int inverse, void* stream) {
    if (!in || !out || B <= 0 || win_h <= 0 || win_w <= 0 || Ph % win_h || Pw % win_w) return DKD_ERR_BAD_ARG;
    if (D & 3) return DKD_ERR_ALIGN;
    const long nvec = (long)B * Ph * Pw * (D / 4);
    hipLaunchKernelGGL(window_permute_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, F4(in), F4W(out), B, Ph,
                       Pw, win_h, win_w, D / 4, inverse, nvec);
    DKD_LAUNCH_CHECK();
    return DKD_OK;
}
