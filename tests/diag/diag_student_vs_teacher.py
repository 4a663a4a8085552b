"""Config [4] race hunt, stage 2: single student-side launches (and short chains of them) repeated on the main stream while the
REAL bf16 ViT-g teacher forward (bs 32) runs on a side stream; reports which launch's output bits depend on the co-runner."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd import ops
from dinov2_distillation_amd.models import DINOv2ViT

torch.manual_seed(0)
dev = "cuda"
prec = os.environ.get("DKD_DIAG_TEACHER_PREC", "bf16")
teacher = DINOv2ViT(os.environ.get("DKD_DIAG_TEACHER", "dinov2_vitg14"), seed=0, precision=None if prec == "f32" else prec).cuda()
xb = torch.randn(32, 3, 224, 224, device=dev)
side = torch.cuda.Stream()
with torch.no_grad():
    teacher(xb)
torch.cuda.synchronize()


def noise():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        teacher(xb)


def check(name, f, reps=6):
    torch.cuda.synchronize()
    ref = [t.clone() for t in f()]
    torch.cuda.synchronize()
    bad, worst, nel = 0, 0.0, 0
    for _ in range(reps):
        noise()
        out = f()
        torch.cuda.synchronize()
        for a, b in zip(out, ref):
            if not torch.equal(a, b):
                bad += 1
                worst = max(worst, ((a - b).abs().max() / b.abs().max()).item())
                nel = max(nel, int((a != b).sum()))
                break
    print(f"{name:70s} {'OK' if bad == 0 else f'DIFFERS in {bad}/{reps} runs, up to {nel} elements, worst rel {worst:.2e}'}", flush=True)


B = 32
a = torch.randn(1568, 4096, device=dev); w = torch.randn(1024, 4096, device=dev) * 0.02; res = torch.randn(1568, 1024, device=dev)
bias = torch.randn(1024, device=dev)
check("gemm f32 1568x1024x4096 bias+residual (stage-4 pwconv2)", lambda: [ops.gemm(a, w, bias=bias, residual=res)])
y = ops.gemm(a, w, bias=bias, residual=res)
check("resize 32x7x7x1024 -> 16x16", lambda: [ops.resize_fwd(y.view(32, 7, 7, 1024), 16, 16)])
check("gemm -> resize chain", lambda: [ops.resize_fwd(ops.gemm(a, w, bias=bias, residual=res).view(32, 7, 7, 1024), 16, 16)])
a3 = torch.randn(6272, 2048, device=dev); w3 = torch.randn(512, 2048, device=dev) * 0.02; r3 = torch.randn(6272, 512, device=dev)
check("gemm f32 6272x512x2048 (stage-3 pwconv2) -> resize 14->16", lambda: [ops.resize_fwd(ops.gemm(a3, w3, residual=r3).view(32, 14, 14, 512), 16, 16)])
x4 = torch.randn(32, 7, 7, 1024, device=dev); wt = torch.randn(49, 1024, device=dev); b4 = torch.randn(1024, device=dev)
check("dwconv7 32x7x7x1024", lambda: [ops.dwconv7(x4, wt, b4)])
x3 = torch.randn(32, 14, 14, 512, device=dev); wt3 = torch.randn(49, 512, device=dev)
check("dwconv7 32x14x14x512", lambda: [ops.dwconv7(x3, wt3, None)])
x1 = torch.randn(32, 56, 56, 128, device=dev); wt1 = torch.randn(49, 128, device=dev)
check("dwconv7 32x56x56x128", lambda: [ops.dwconv7(x1, wt1, None)])
g4 = torch.randn(1024, device=dev)
check("layernorm 1568x1024", lambda: list(ops.layernorm_fwd(x4.view(1568, 1024), g4, b4, 1e-6)))
h = torch.randn(1568, 4096, device=dev); gg = torch.randn(4096, device=dev); gb = torch.randn(4096, device=dev)
check("grn_fwd (preact) 1568x4096", lambda: [ops.grn_fwd(h, gg, gb, 32, preact=True)[0]])
a1 = torch.randn(1568, 1024, device=dev); w1 = torch.randn(4096, 1024, device=dev) * 0.02
check("gemm f32 1568x4096x1024 (stage-4 pwconv1)", lambda: [ops.gemm(a1, w1)])
aa = torch.randn(100352, 128, device=dev); ww = torch.randn(512, 128, device=dev)
check("gemm f32 100352x512x128 (stage-1 pwconv1)", lambda: [ops.gemm(aa, ww)])
