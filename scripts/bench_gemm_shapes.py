"""GPU microbenchmark: every distinct GEMM shape of one config-2 step (bs=64), TFLOP/s per shape and layout."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
from dinov2_distillation_amd.trainer import DistillationStepper

shapes = collections.Counter()
inner = ops.gemm
def rec(a, b, **kw):
    al, bl = kw.get("a_layout", 0), kw.get("b_layout", 0)
    a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    M, K = (a2.shape[0], a2.shape[1]) if al == 0 else (a2.shape[1], a2.shape[0])
    N = b2.shape[0] if bl == 0 else b2.shape[1]
    shapes[(M, N, K, al, bl)] += 1
    return inner(a, b, **kw)
ops.gemm = rec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = DistillationStepper()
x = torch.randn(B, 3, 224, 224).cuda()
st.step(x)
torch.cuda.synchronize()
ops.gemm = inner
rows = []
for (M, N, K, al, bl), cnt in shapes.items():
    a = torch.randn((M, K) if al == 0 else (K, M), device="cuda")
    b = torch.randn((N, K) if bl == 0 else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(2): inner(a, b, a_layout=al, b_layout=bl, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): inner(a, b, a_layout=al, b_layout=bl, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * N * K
    # vendor library (rocBLAS / hipBLASLt through torch.matmul, strict fp32) on the same operands: a yardstick only
    torch.backends.cuda.matmul.allow_tf32 = False
    am = a if al == 0 else a.t()
    bm = b.t() if bl == 0 else b
    for _ in range(2): torch.matmul(am, bm, out=out)
    e0.record()
    for _ in range(reps): torch.matmul(am, bm, out=out)
    e1.record(); torch.cuda.synchronize()
    us_lib = e0.elapsed_time(e1) * 1e3 / reps
    rows.append((cnt * us, cnt, M, N, K, al, bl, us, fl / us / 1e6, us_lib, fl / us_lib / 1e6))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total gemm time per step: %.2f ms over %d launches" % (tot / 1e3, sum(r[1] for r in rows)))
for r in rows[:40]:
    print("%7.1f us total  x%3d  M=%6d N=%5d K=%6d a%d b%d  %8.1f us  %6.1f TF/s | torch.matmul %8.1f us %6.1f TF/s" % r)
print("vendor-library total for the same launches: %.2f ms" % (sum(r[1] * r[9] for r in rows) / 1e3))
