"""roctx ranges around every autograd node of the step (the reference relies on Lightning's profiler hooks; here a
rocprofv3 --marker-trace / --kernel-trace run is readable by op instead of by mangled kernel name).

    DKD_ROCTX=1 rocprofv3 --marker-trace --kernel-trace ... -- python3 bench.py ...

Off by default (zero cost: nothing is wrapped).  ``torch.cuda.nvtx`` IS roctx on a ROCm build of PyTorch."""
import functools
import os

import torch

ENABLED = os.environ.get("DKD_ROCTX", "0") == "1"


def _wrap(fn, label):
    @functools.wraps(fn)
    def inner(*a, **k):
        torch.cuda.nvtx.range_push(label)
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.nvtx.range_pop()
    return inner


def instrument(namespace):
    """wrap forward / backward of every torch.autograd.Function defined in the given module namespace"""
    if not ENABLED:
        return 0
    n = 0
    for name, obj in list(namespace.items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function \
                and obj.__module__ == namespace.get("__name__"):
            for meth in ("forward", "backward"):
                raw = obj.__dict__.get(meth)
                if raw is not None:
                    f = raw.__func__ if isinstance(raw, staticmethod) else raw
                    setattr(obj, meth, staticmethod(_wrap(f, f"dkd/{name}.{meth}")))
                    n += 1
    return n
