"""How far ahead of the GPU is the host?  Enqueue time of K steps (no sync) vs their GPU completion time, for both bindings of
the same libdkd_hip.so (DKD_BINDING=ctypes: raw C ABI through ctypes; torch: torch.ops.dkd.*, TORCH_LIBRARY) and for the
launch-bound BASELINE configs[0] (bs 2) next to configs[1] (bs 64).
    python scripts/host_overhead.py            # spawns itself once per binding"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "DKD_HOST_OVERHEAD_CHILD" not in os.environ:
    for b in ("ctypes", "torch"):
        env = dict(os.environ, DKD_BINDING=b, DKD_HOST_OVERHEAD_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from dinov2_distillation_amd import _lib
from dinov2_distillation_amd.trainer import DistillationStepper
for student, bs in (("convnext_tiny", 64), ("stdc_2", 2), ("convnext_tiny", 2)):
    st = DistillationStepper(student=student, device="cuda:0")
    x = torch.randn(bs, 3, 224, 224).cuda()
    for _ in range(3): st.step(x)
    torch.cuda.synchronize()
    for K in (10,):
        t0 = time.perf_counter()
        for _ in range(K): st.step(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"binding={_lib.BINDING:6s} vits14->{student} bs={bs}: host enqueue {1e3*(t1-t0)/K:.1f} ms/step, wall {1e3*(t2-t0)/K:.1f} ms/step", flush=True)
    del st
