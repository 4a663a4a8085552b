"""bf16 GEMM: the LDS-DMA ring kernel (csrc/gemm_bf16_ring.hip) against the register-staged kernel (variant 1) - bit equality in
every layout / epilogue - and timings of every ring configuration beside the old kernel and the vendor library on the step's shapes.
    python3 scripts/tune_bf16_ring.py [check|bench|all] [cfgs...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
cfgs = [int(c) for c in sys.argv[2:]] or [11, 12, 13, 14, 15, 16]
dev = "cuda"


def t(f, reps=20):
    try:
        f()
    except Exception:  # a configuration that does not support this layout / shape
        return float("nan")
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


def operands(M, N, K, al, bl, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = (torch.randn((M, K) if al == 0 else (K, M), device=dev, generator=g)).bfloat16()
    b = (torch.randn((N, K) if bl == 0 else (K, N), device=dev, generator=g) * K ** -0.5).bfloat16()
    return a, b


if mode in ("check", "all"):
    bad = 0
    cases = [(16448, 384, 384), (12544, 384, 1536), (3136, 768, 3072), (1000, 200, 192), (129, 136, 64), (50176, 192, 768)]
    for (M, N, K) in cases:
        for (al, bl) in ((0, 0), (0, 1)):
            a, b = operands(M, N, K, al, bl)
            bias = torch.randn(N, device=dev).bfloat16()
            res = torch.randn(M, N, device=dev)
            cs = torch.rand(N, device=dev) + 0.5
            r16 = torch.randn(M, N, device=dev).bfloat16()
            for name, kw in (("bf16+bias", dict(epi=ops.EPI_BF16, bias=bias)), ("gelu", dict(epi=ops.EPI_BF16_GELU, bias=bias)),
                             ("relu", dict(epi=ops.EPI_BF16_RELU, bias=bias)), ("f32", dict(epi=ops.EPI_F32)),
                             ("f32 resid", dict(epi=ops.EPI_F32_RESID, bias=bias, colscale=cs, residual=res))):
                ref = ops.gemm_bf16(a, b, a_layout=al, b_layout=bl, variant=1, **kw)
                for c in cfgs:
                    out = ops.gemm_bf16(a, b, a_layout=al, b_layout=bl, variant=c, **kw)
                    if not torch.equal(out, ref):
                        d = (out.float() - ref.float()).abs()
                        print(f"MISMATCH {M}x{N}x{K} a{al} b{bl} {name} cfg {c}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")
                        bad += 1
            ref = ops.gemm_bf16_resid16(a, b, bias, r16) if (al, bl) == (0, 0) else None
    # weight-gradient layout with the fused bias gradient
    for (M, N, K) in [(1536, 384, 12544), (384, 1536, 12544), (96, 384, 200704), (768, 3072, 3136), (384, 384, 16384), (136, 200, 8192)]:
        a, b = operands(M, N, K, 1, 1)
        ref = ops.gemm_bf16(a, b, a_layout=1, b_layout=1, epi=ops.EPI_F32, variant=1)
        rs_ref = ops.colreduce16(a, ops.RED_SUM, round_out=True).reshape(-1)
        for c in cfgs:
            rs = torch.empty(M, device=dev)
            try:
                out = ops.gemm_bf16(a, b, a_layout=1, b_layout=1, epi=ops.EPI_F32, variant=c, rowsum=rs)
            except Exception:
                continue
            out2 = ops.gemm_bf16(a, b, a_layout=1, b_layout=1, epi=ops.EPI_F32, variant=c)
            e = float((out - ref).abs().max() / ref.abs().max())
            e2 = float((rs - rs_ref).abs().max() / rs_ref.abs().max())
            ok = torch.equal(out, ref) and torch.equal(out2, ref)
            # (row sums: fp32 sums in a different order, then one bf16 rounding: at most one bf16 ulp apart)
            if not ok or e2 > 2 ** -7:
                print(f"wgrad {M}x{N}x{K} cfg {c}: dW equal={ok} rel err {e:.2e}; rowsum rel err {e2:.2e}")
                bad += not ok or e2 > 2 ** -7
    print("check:", "all equal" if not bad else f"{bad} FAILURES", flush=True)

if mode in ("bench", "all"):
    fwd = [(16448, 1152, 384), (16448, 1536, 384), (16448, 384, 1536), (16448, 384, 384), (50176, 768, 192), (50176, 192, 768),
           (12544, 1536, 384), (12544, 384, 1536), (3136, 3072, 768), (3136, 768, 3072), (16384, 384, 384), (16384, 1536, 384),
           (16384, 384, 1536), (16384, 384, 768), (12544, 384, 768), (3136, 768, 1536), (50176, 192, 384)]
    hdr = "shape".ljust(34) + " old   " + " ".join(f"c{c:<4d}" for c in cfgs) + " vendor  floor"
    for title, layouts in (("forward (NT)", (0, 0)), ("dgrad (NN)", (0, 1))):
        print(f"--- {title}: us per launch\n{hdr}")
        for (M, N, K) in fwd:
            if layouts == (0, 1):
                N, K = K, N  # dx[M, K_in] = dy[M, N_out] W[N_out, K_in]: reduction over the forward's N
            a, b = operands(M, N, K, *layouts)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            row = [t(lambda: ops.gemm_bf16(a, b, a_layout=layouts[0], b_layout=layouts[1], out=out, variant=1))]
            for c in cfgs:
                row.append(t(lambda: ops.gemm_bf16(a, b, a_layout=layouts[0], b_layout=layouts[1], out=out, variant=c)))
            bm = b.t() if layouts[1] == 0 else b
            v = t(lambda: torch.matmul(a, bm, out=out))
            by = (M * K + N * K + M * N) * 2
            print(f"{M:7d} x {N:5d} x {K:5d} a{layouts[0]}b{layouts[1]}".ljust(34) + " ".join(f"{x:5.1f}" for x in row) + f" {v:6.1f} {by / 6.0e6:6.1f}", flush=True)
    print(f"--- weight gradient (TN, split-K + fold; rs = with the fused bias gradient)\n{hdr}")
    for (M, N, K) in [(1152, 384, 16448), (1536, 384, 16384), (384, 1536, 16384), (384, 384, 16384), (384, 96, 200704), (96, 384, 200704),
                      (768, 192, 50176), (192, 768, 50176), (1536, 384, 12544), (384, 1536, 12544), (3072, 768, 3136), (768, 3072, 3136)]:
        a, b = operands(M, N, K, 1, 1)
        out = torch.empty(M, N, device=dev)
        rs = torch.empty(M, device=dev)
        row = [t(lambda: ops.gemm_bf16(a, b, a_layout=1, b_layout=1, epi=ops.EPI_F32, out=out, variant=1))]
        for c in cfgs:
            row.append(t(lambda: ops.gemm_bf16(a, b, a_layout=1, b_layout=1, epi=ops.EPI_F32, out=out, variant=c, rowsum=rs)))
        v = t(lambda: torch.matmul(a.t(), b))
        print(f"{M:7d} x {N:5d} x {K:6d} a1b1".ljust(34) + " ".join(f"{x:5.1f}" for x in row) + f" {v:6.1f}", flush=True)
