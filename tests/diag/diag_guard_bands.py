"""Out-of-bounds write hunt: every output the ops layer allocates (torch.empty / empty_like inside dinov2_distillation_amd.ops
and .functional) is carved out of the middle of a larger sentinel-filled buffer; after the pass the guard bands on both sides
of every output must still hold the sentinel.  Runs the teacher forward (and optionally a whole training step).
    python tests/diag/diag_guard_bands.py [teacher] [precision] [batch] [step]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd import functional as Fn, ops
from dinov2_distillation_amd.models import DINOv2ViT

SENT = 1234.5
BAND = int(os.environ.get("DKD_GUARD_BAND", str(1 << 20)))  # elements on each side
records = []


class Proxy:
    def __getattr__(self, k):
        return getattr(torch, k)

    def _alloc(self, shape, dtype, device, what):
        n = 1
        for s in shape:
            n *= int(s)
        if n == 0 or torch.device(device).type != "cuda":
            return torch.empty(shape, dtype=dtype, device=device)
        buf = torch.full((n + 2 * BAND,), SENT if dtype.is_floating_point else 77, dtype=dtype, device=device)
        view = buf[BAND:BAND + n].view(shape)
        import traceback
        fr = [f for f in traceback.extract_stack(limit=8) if "diag_guard" not in f.filename][-1]
        records.append((buf, n, f"{what}{tuple(shape)} at {os.path.basename(fr.filename)}:{fr.lineno} ({fr.name})"))
        return view

    def empty(self, *shape, dtype=torch.float32, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        return self._alloc(shape, dtype, device, "empty")

    def empty_like(self, t, **kw):
        return self._alloc(tuple(t.shape), t.dtype, t.device, "empty_like")


ops.torch = Proxy()
Fn.torch = Proxy()
name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitg14"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
teacher = DINOv2ViT(name, seed=0, precision=None if prec == "f32" else prec).cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
with torch.no_grad():
    out = teacher(x)["feature_map"]
torch.cuda.synchronize()
bad = 0
for buf, n, what in records:
    lo, hi = buf[:BAND], buf[BAND + n:]
    sent = SENT if buf.dtype.is_floating_point else 77
    nlo, nhi = int((lo != sent).sum()), int((hi != sent).sum())
    if nlo or nhi:
        bad += 1
        il = (lo != sent).nonzero().reshape(-1)
        ih = (hi != sent).nonzero().reshape(-1)
        print(f"OUT-OF-BOUNDS WRITE around {what}: {nlo} elements BEFORE (offsets {[int(v) - BAND for v in il[:4]]} .. {[int(v) - BAND for v in il[-2:]]}), "
              f"{nhi} elements AFTER (offsets +{[int(v) for v in ih[:4]]} .. +{[int(v) for v in ih[-2:]]}); sample values {hi[ih[:3]].tolist() if nhi else lo[il[:3]].tolist()}")
print(f"{name} {prec} bs {B}: {len(records)} guarded outputs, {bad} with a damaged guard band; feature map finite: {bool(torch.isfinite(out).all())}")
