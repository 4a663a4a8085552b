"""ScaleKD loss-tail kernels (csrc/loss.hip) against the 8 TB/s roofline: forward (reads S and T once) and backward (+ writes dS) at
the shapes of the BASELINE configs, spatial and frequency variants.  "hot": the same buffers every launch (50 MB: they stay in
the 256 MiB Infinity Cache, as they do in the step, where the producer kernel has just written them); "cold": a ring of buffers
larger than the cache.  DKD_HIP_LIB selects another build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops


def timeit(fs, reps=20):
    for f in fs: f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps): fs[i % len(fs)]()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[3]


for (B, N, D) in [(64, 256, 384), (128, 256, 768), (16, 1369, 1024), (32, 256, 1536)]:
    rows = B * N
    nb = max(2, int(600e6 / (rows * D * 8)) + 1)
    S = [torch.randn(B, N, D, device="cuda") for _ in range(nb)]
    T = [torch.randn(B, N, D, device="cuda") for _ in range(nb)]
    g = torch.ones(1, device="cuda")
    for freq in (False, True):
        ms = ops.colreduce(S[0], groups=B, scale=1.0 / N) if freq else None
        mt = ops.colreduce(T[0], groups=B, scale=1.0 / N) if freq else None
        out = []
        for name, nbuf, mk in (("fwd", 2, lambda i: (lambda: ops.kd_loss_fwd(S[i], T[i], N, 0.08, ms, mt))),
                               ("bwd", 3, lambda i: (lambda: ops.kd_loss_bwd(S[i], T[i], N, 0.08, g, ms, mt)))):
            hot = timeit([mk(0)])
            cold = timeit([mk(i) for i in range(nb)])
            by = nbuf * rows * D * 4
            out.append(f"{name} hot {hot:6.1f} us {by / hot * 1e-6:5.2f} TB/s ({by / hot * 1e-6 / 8:.2f}) | cold {cold:6.1f} us "
                       f"{by / cold * 1e-6:5.2f} TB/s ({by / cold * 1e-6 / 8:.2f})")
        print(f"B={B} N={N} D={D} {'freq' if freq else 'spat'}: " + " || ".join(out), flush=True)
