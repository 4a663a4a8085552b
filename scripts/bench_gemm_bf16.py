"""bf16-in-HBM GEMM (csrc/gemm_bf16.hip) on the teacher's shapes: TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak, beside the
fp32-activation / bf16-operand kernel it replaces in the autocast teacher and the exact fp32 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


for (M, N, K, what) in [(16448, 1152, 384, "vits qkv"), (16448, 1536, 384, "vits fc1"), (16448, 384, 1536, "vits fc2"),
                        (32896, 2304, 768, "vitb qkv bs128"), (21920, 4096, 1024, "vitl@518 fc1 bs16"), (21920, 1024, 4096, "vitl@518 fc2"),
                        (8224, 4608, 1536, "vitg qkv bs32"), (8224, 8192, 1536, "vitg w12"), (8224, 1536, 4096, "vitg w3")]:
    a32 = torch.randn(M, K, device="cuda"); w32 = torch.randn(N, K, device="cuda") * K ** -0.5
    a16, w16 = a32.bfloat16(), w32.bfloat16()
    res = torch.randn(M, N, device="cuda"); ls = torch.rand(N, device="cuda"); b16 = torch.randn(N, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    u0 = t(lambda: ops.gemm_bf16(a16, w16, epi=ops.EPI_BF16, bias=b16))
    u1 = t(lambda: ops.gemm_bf16(a16, w16, epi=ops.EPI_F32_RESID, bias=b16, colscale=ls, residual=res))
    u2 = t(lambda: ops.gemm(a32, w32, precision=ops.PREC_BF16))
    u3 = t(lambda: ops.gemm(a32, w32))
    print(f"{what:20s} {M}x{N}x{K}: bf16/bf16-out {u0:7.1f} us {fl / u0 * 1e-6:6.0f} TF/s ({fl / u0 * 1e-6 / 2500:.2f} of 2.5 PF) | "
          f"f32-out+resid {u1:7.1f} us {fl / u1 * 1e-6:6.0f} TF/s | fp32 act, bf16 operands {u2:7.1f} us {fl / u2 * 1e-6:6.0f} TF/s | "
          f"exact fp32 {u3:7.1f} us {fl / u3 * 1e-6:6.0f} TF/s", flush=True)

# the trainable half's backward matmuls (ConvNeXt-tiny pointwise convs, bs 64) from the operands as they lie
print("--- dgrad (dy[M,N] . W[N,K]) and weight gradient (dy^T x, K-slabs) ---", flush=True)
for (rows, n_out, k_in, what) in [(200704, 384, 96, "stage1 pw1"), (200704, 96, 384, "stage1 pw2"), (50176, 768, 192, "stage2 pw1"),
                                  (12544, 1536, 384, "stage3 pw1"), (12544, 384, 1536, "stage3 pw2"), (3136, 3072, 768, "stage4 pw1"),
                                  (16384, 384, 384, "projector q/k/v")]:
    x = torch.randn(rows, k_in, device="cuda").bfloat16(); w = (torch.randn(n_out, k_in, device="cuda") * k_in ** -0.5).bfloat16()
    dy = torch.randn(rows, n_out, device="cuda").bfloat16()
    x32, w32, dy32 = x.float(), w.float(), dy.float()
    fl = 2.0 * rows * n_out * k_in
    u_f = t(lambda: ops.gemm_bf16(x, w))
    u_d = t(lambda: ops.gemm_bf16(dy, w, b_layout=1))
    u_w = t(lambda: ops.gemm_bf16(dy, x, a_layout=1, b_layout=1, epi=ops.EPI_F32))
    u_d32 = t(lambda: ops.gemm(dy32, w32, b_layout=1))
    u_w32 = t(lambda: ops.gemm(dy32, x32, a_layout=1, b_layout=1))
    print(f"{what:16s} rows {rows} out {n_out} in {k_in}: fwd {u_f:6.1f} us {fl / u_f * 1e-6:5.0f} TF/s | dgrad {u_d:6.1f} us {fl / u_d * 1e-6:5.0f} TF/s "
          f"(fp32 {u_d32:6.1f} us) | wgrad {u_w:6.1f} us {fl / u_w * 1e-6:5.0f} TF/s (fp32 {u_w32:6.1f} us)", flush=True)
