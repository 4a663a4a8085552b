"""cProfile of the host side of one step (main thread: forward + optimiser; the backward runs on autograd's own thread and is
profiled through a hook that installs the profiler there).  python3 scripts/host_profile.py [precision|none]"""
import cProfile, io, os, pstats, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd.trainer import DistillationStepper

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16_autocast"
prec = None if prec == "none" else prec
st = DistillationStepper(device="cuda:0", precision=prec)
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3):
    st.step(x)
torch.cuda.synchronize()
main_prof, bwd_prof = cProfile.Profile(), cProfile.Profile()
bwd_thread = {}


class Tap(torch.autograd.Function):  # first node of the backward pass: runs on the autograd thread
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if "on" not in bwd_thread:
            bwd_thread["on"] = threading.get_ident()
            bwd_prof.enable()
        return g


K = 5
main_prof.enable()
for _ in range(K):
    st.optimizer.zero_grad()
    loss = Tap.apply(st.module.training_step(x, 0))
    loss.backward()
    st.optimizer.step(clip_grad_norm=1.0)
main_prof.disable()
torch.cuda.synchronize()
for name, p in (("main thread (forward, optimiser)", main_prof), ("autograd thread (backward)", bwd_prof)):
    try:
        p.disable()
    except Exception:
        pass
    s = io.StringIO()
    ps = pstats.Stats(p, stream=s).sort_stats("tottime")
    ps.print_stats(28)
    print(f"===== {name}: {K} steps\n" + "\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
