"""A/B of the v2 GEMM family (16x16x4 MFMA, K-contiguous LDS rows) against the automatic configuration, per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops

torch.backends.cuda.matmul.allow_tf32 = False
shapes = [  # (M, N, K, a_layout, b_layout, count per step)
    (16448, 384, 1536, 0, 0, 12), (16448, 1536, 384, 0, 0, 12), (16448, 1152, 384, 0, 0, 12), (16384, 1536, 384, 0, 1, 8),
    (16384, 384, 1536, 0, 1, 8), (12544, 384, 1536, 0, 1, 9), (12544, 384, 1536, 0, 0, 9), (16384, 384, 1536, 0, 0, 8),
    (12544, 1536, 384, 0, 1, 9), (1536, 384, 12544, 1, 1, 9), (384, 1536, 12544, 1, 1, 9), (16384, 1536, 384, 0, 0, 8),
    (12544, 1536, 384, 0, 0, 9), (16384, 384, 384, 0, 1, 20), (384, 384, 16384, 1, 1, 16), (16384, 384, 384, 0, 0, 20),
    (16448, 384, 384, 0, 0, 12), (200704, 384, 96, 0, 1, 3), (50176, 192, 768, 0, 1, 3), (200704, 96, 384, 0, 0, 3),
    (3136, 768, 3072, 0, 0, 3), (3136, 3072, 768, 0, 0, 3), (50176, 768, 192, 0, 0, 3),
    (384, 96, 200704, 1, 1, 3), (96, 384, 200704, 1, 1, 3), (768, 3072, 3136, 1, 1, 3), (3072, 768, 3136, 1, 1, 3),
    (768, 192, 50176, 1, 1, 3), (192, 768, 50176, 1, 1, 3), (1536, 384, 16384, 1, 1, 4), (384, 1536, 16384, 1, 1, 4),
    (200704, 384, 96, 0, 0, 3), (200704, 96, 384, 0, 1, 3), (50176, 192, 768, 0, 0, 3), (50176, 768, 192, 0, 1, 3),
    (3136, 768, 3072, 0, 1, 3), (3136, 3072, 768, 0, 1, 3), (16384, 384, 1152, 0, 1, 4), (16384, 1152, 384, 0, 0, 4),
]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3, 44]
tot = {v: 0.0 for v in variants}
best_tot = 0.0
for M, N, K, al, bl, cnt in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((M, K) if al == 0 else (K, M), generator=g).cuda()
    b = torch.randn((N, K) if bl == 0 else (K, N), generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    ref = None
    line = "M=%6d N=%5d K=%6d a%d b%d x%2d |" % (M, N, K, al, bl, cnt)
    times, outs, kws = {}, {}, {}
    for v in variants:
        out = torch.empty(M, N, device="cuda")
        kw = dict(a_layout=al, b_layout=bl, out=out, variant=v)
        if al == 0:
            kw.update(bias=bias, residual=res)
        try:
            for _ in range(2): ops.gemm(a, b, **kw)
        except Exception as e:
            line += " v%d: %s |" % (v, str(e)[-24:])
            continue
        kws[v], outs[v] = kw, out
    torch.cuda.synchronize()
    samples = {v: [] for v in kws}
    for rnd in range(7):  # interleaved rounds: clock / thermal drift hits every variant alike; report the median
        for v in (list(kws) if rnd % 2 == 0 else list(kws)[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm(a, b, **kws[v])
            e1.record(); torch.cuda.synchronize()
            samples[v].append(e0.elapsed_time(e1) * 1e3 / 3)
    for v in kws:
        us = sorted(samples[v])[len(samples[v]) // 2]
        if ref is None:
            ref = outs[v].clone()
            err = 0.0
        else:
            err = ((outs[v] - ref).abs().max() / ref.abs().max()).item()
        times[v] = us
        tot[v] += cnt * us
        line += " v%d %6.1fus %5.1fTF%s |" % (v, us, 2.0 * M * N * K / us / 1e6, "" if err < 1e-5 else " ERR %.1e" % err)
    best_tot += cnt * min(times.values())
    print(line, flush=True)
print("per-step totals (ms):", {v: round(t / 1e3, 2) for v, t in tot.items()}, "best-of per shape: %.2f" % (best_tot / 1e3))
