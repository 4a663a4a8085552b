"""At the end of backward: how far behind the main stream are the producer side streams (weight-gradient stream, frequency-chain
stream)?  If a side stream finishes later than the main chain, IT bounds the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
from dinov2_distillation_amd.trainer import DistillationStepper
st = DistillationStepper(device="cuda:0")
x = torch.randn(64, 3, 224, 224).cuda()
for _ in range(3): st.step(x)
torch.cuda.synchronize()
lags = []
for _ in range(6):
    st.optimizer.zero_grad()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    loss = st.module.training_step(x, 0)
    loss.backward()
    main = torch.cuda.current_stream()
    em = torch.cuda.Event(enable_timing=True); em.record(main)
    evs = []
    for s in ops.producer_streams():
        e = torch.cuda.Event(enable_timing=True); e.record(s); evs.append(e)
    st.optimizer.step(clip_grad_norm=st.clip)
    torch.cuda.synchronize()
    lags.append((e0.elapsed_time(em), [e0.elapsed_time(e) for e in evs]))
for m, ss in lags:
    print("main chain done at %.2f ms; producer streams done at %s ms" % (m, ", ".join("%.2f" % v for v in ss)))
