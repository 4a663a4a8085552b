"""Fused AdamW over one flat HBM arena (params | grads | exp_avg | exp_avg_sq), drop-in for the
``torch.optim.AdamW`` the reference builds in ``configure_optimizers`` (train/distillation_module.py:457-479): same
param-group interface (schedulers mutate ``group['lr']``), same update rule, plus an optional fused clip-by-global-norm
(Lightning's ``gradient_clip_val``, train.py:267-268) that needs no host synchronisation.

Every parameter becomes a view into the arena and owns a persistent ``.grad`` view, so (i) autograd accumulates
straight into the arena, (ii) the DDP all-reduce works on contiguous bucket slices without flatten/unflatten copies,
(iii) the whole optimiser is 2 + n_groups kernel launches.
"""
import ctypes
import os

import torch

from . import _lib, ops
from .ops import _p, _stream, workspace

ALIGN = 64  # elements: keep every tensor 256-byte aligned inside the arena


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay))
        super().__init__(params, defaults)
        self._build_arena()
        self._steps = 0
        self.graph_safe = False  # True: every optimiser scalar lives in device memory (hipGraph capture of the step)
        self._hyper = None
        # weight gradients may now be written into the arena from the side stream (DKD_ASYNC_WGRAD=0 keeps one stream)
        ops.ASYNC_WGRAD = self.flat_g.is_cuda and os.environ.get("DKD_ASYNC_WGRAD", "1") != "0"

    def _build_arena(self):
        plist = []
        for g in self.param_groups:
            g["_range"] = None
            start = sum(n for _, _, n in plist)
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                n = (p.numel() + ALIGN - 1) // ALIGN * ALIGN
                plist.append((p, sum(m for _, _, m in plist), n))
            g["_range"] = (start, sum(n for _, _, n in plist))
        if not plist:
            raise ValueError("FusedAdamW: no trainable parameters")
        dev = plist[0][0].device
        total = sum(n for _, _, n in plist)
        self.arena = torch.zeros((4, total), dtype=torch.float32, device=dev)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = self.arena[0], self.arena[1], self.arena[2], self.arena[3]
        self.layout = []
        with torch.no_grad():
            for p, off, n in plist:
                if p.dtype != torch.float32 or p.device != dev:
                    raise ValueError("FusedAdamW: all parameters must be fp32 on one device")
                view = self.flat_p[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
                self.layout.append((p, off, p.numel()))
        self.total = total
        self._gnorm = torch.zeros(2, dtype=torch.float32, device=dev)

    def sync_hyper(self):
        """(graph-safe mode) push lr / betas / eps / weight_decay of every param group to the device; the step counter is kept.
        Call after a scheduler changed ``group['lr']`` - it is a small H2D copy outside the captured graph."""
        rows = [[g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], 0.0] for g in self.param_groups]
        host = torch.tensor(rows, dtype=torch.float32)
        if self._hyper is None:
            host[:, 5] = float(self._steps)  # completed steps so far; step() increments it on the device before use
            self._hyper = host.to(self.flat_p.device)
        else:
            self._hyper[:, :5].copy_(host[:, :5], non_blocking=True)

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p, off, n in self.layout:  # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + n].view(p.shape)

    def grad_norm(self):
        """global L2 norm of all gradients, left on the device: tensor [sum_sq, norm]"""
        lib = _lib.load()
        if not self.flat_g.is_cuda:
            raise _lib.DkdError("FusedAdamW needs parameters on the GPU (no CPU path)")
        ops.wgrad_join()
        ws = workspace(lib.dkd_sumsq_f32_workspace_bytes(self.total), self.flat_g.device, slot=2)
        _lib.check(lib.dkd_sumsq_f32(_p(self.flat_g), self.total, _p(self._gnorm), _p(ws), _stream()), "dkd_sumsq_f32")
        return self._gnorm

    @torch.no_grad()
    def step(self, closure=None, clip_grad_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        ops.wgrad_join()  # the side-stream weight gradients must have landed in the arena
        gn = self.grad_norm() if clip_grad_norm is not None else None
        if self.graph_safe and self._hyper is None:
            self.sync_hyper()
        self._steps += 1
        if self.graph_safe:
            self._hyper[:, 5] += 1.0  # the step counter advances on the device (captured with the graph)
            for i, g in enumerate(self.param_groups):
                a, b = g["_range"]
                if b > a:
                    _lib.check(lib.dkd_adamw_dev_f32(ctypes.c_void_p(self.flat_p.data_ptr() + 4 * a),
                                                     ctypes.c_void_p(self.flat_g.data_ptr() + 4 * a),
                                                     ctypes.c_void_p(self.flat_m.data_ptr() + 4 * a),
                                                     ctypes.c_void_p(self.flat_v.data_ptr() + 4 * a), b - a,
                                                     ctypes.c_void_p(self._hyper.data_ptr() + 24 * i), _p(gn),
                                                     float(clip_grad_norm or 0.0), _stream()), "dkd_adamw_dev_f32")
            return loss
        for g in self.param_groups:
            a, b = g["_range"]
            if b <= a:
                continue
            _lib.check(lib.dkd_adamw_f32(ctypes.c_void_p(self.flat_p.data_ptr() + 4 * a),
                                         ctypes.c_void_p(self.flat_g.data_ptr() + 4 * a),
                                         ctypes.c_void_p(self.flat_m.data_ptr() + 4 * a),
                                         ctypes.c_void_p(self.flat_v.data_ptr() + 4 * a), b - a, g["lr"], g["betas"][0],
                                         g["betas"][1], g["eps"], g["weight_decay"], self._steps, _p(gn),
                                         float(clip_grad_norm or 0.0), _stream()), "dkd_adamw_f32")
        return loss
