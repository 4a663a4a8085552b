"""GEMM variants on the large-model shapes (vitl14 @518, vitg14): does the 64x64 default still win?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
shapes = [(21920, 3072, 1024, 0, 0), (21920, 1024, 1024, 0, 0), (21920, 4096, 1024, 0, 0), (21920, 1024, 4096, 0, 0),
          (8224, 4608, 1536, 0, 0), (8224, 1536, 1536, 0, 0), (8224, 8192, 1536, 0, 0), (8224, 1536, 4096, 0, 0),
          (21904, 1024, 4096, 0, 1), (21904, 4096, 1024, 0, 1), (1024, 4096, 21904, 1, 1), (16448, 768, 3072, 0, 0), (16448, 2304, 768, 0, 0)]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 44, 101, 100]
for M, N, K, al, bl in shapes:
    a = torch.randn((M, K) if al == 0 else (K, M), device="cuda")
    b = torch.randn((N, K) if bl == 0 else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    line = "M=%6d N=%5d K=%6d a%d b%d |" % (M, N, K, al, bl)
    samples = {v: [] for v in variants}
    for v in variants:
        for _ in range(2): ops.gemm(a, b, a_layout=al, b_layout=bl, out=out, variant=v)
    for rnd in range(5):
        for v in (variants if rnd % 2 == 0 else variants[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2): ops.gemm(a, b, a_layout=al, b_layout=bl, out=out, variant=v)
            e1.record(); torch.cuda.synchronize()
            samples[v].append(e0.elapsed_time(e1) * 1e3 / 2)
    for v in variants:
        us = sorted(samples[v])[2]
        line += " v%d %7.1fus %5.1fTF |" % (v, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
