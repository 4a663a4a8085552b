"""DINOv2 ViT teacher on MI355X kernels - same module surface as /root/reference/models/backbones/dinov2.py:
``DINOv2ViT(model_name)``, ``forward(x) -> {'feature_map': [B, D, P, P]}`` (a permuted view of token-major memory,
dinov2.py:40), ``.model.blocks[i]`` callable ``[B,N,D] -> [B,N,D]`` with autograd to the input
(train/distillation_module.py:169-177).

The reference pulls architecture + weights from torch.hub (dinov2.py:20); there is no network here, so the published
architecture is built locally with synthetic (seeded) weights under the public checkpoint's parameter names -
``load_state_dict`` of a real DINOv2 checkpoint therefore works unchanged.  Teacher parity vs upstream is unpinned
(see DESIGN.md).
"""
import math
import os

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops

VIT_CONFIGS = {  # embed dims: train.py:103-108; depth / heads: published models (head_dim 64 everywhere)
    "dinov2_vits14": dict(embed_dim=384, depth=12, num_heads=6, ffn="mlp"),
    "dinov2_vitb14": dict(embed_dim=768, depth=12, num_heads=12, ffn="mlp"),
    "dinov2_vitl14": dict(embed_dim=1024, depth=24, num_heads=16, ffn="mlp"),
    "dinov2_vitg14": dict(embed_dim=1536, depth=40, num_heads=24, ffn="swiglu"),
}
PATCH, POS_GRID = 14, 37
# the autocast teacher's attention: bf16-MFMA flash kernel (csrc/attention16_flash.hip); 0 = the fp32-MFMA kernel's round_bf16 mode (A/B)
FLASH16 = os.environ.get("DKD_FLASH16", "1") != "0"
# the autocast teacher's forward as one hipGraph launch after two eager calls per input shape (VisionTransformer._forward_tokens_autocast_graph)
TEACHER_GRAPH = os.environ.get("DKD_TEACHER_GRAPH", "1") != "0"
TEACHER_GRAPH_SHAPES = 4   # input shapes that keep a captured graph (and its private memory pool) alive at a time


def _tn(*shape, gen):
    t = torch.empty(*shape)
    nn.init.trunc_normal_(t, std=0.02, generator=gen)
    return nn.Parameter(t, requires_grad=False)


class _Holder(nn.Module):
    pass


class Block(nn.Module):
    """Parameter names of a DINOv2 block: norm1, attn.{qkv,proj}, ls1.gamma, norm2, mlp.{fc1,fc2}, ls2.gamma."""

    def __init__(self, dim, num_heads, gen, ffn="mlp"):
        super().__init__()
        self.num_heads, self.ffn = num_heads, ffn
        z = lambda n: nn.Parameter(torch.zeros(n), requires_grad=False)
        o = lambda n: nn.Parameter(torch.ones(n), requires_grad=False)
        self.norm1, self.norm2, self.attn, self.mlp, self.ls1, self.ls2 = (_Holder() for _ in range(6))
        self.norm1.weight, self.norm1.bias = o(dim), z(dim)
        self.norm2.weight, self.norm2.bias = o(dim), z(dim)
        self.attn.qkv, self.attn.proj, self.mlp.fc1, self.mlp.fc2 = (_Holder() for _ in range(4))
        self.attn.qkv.weight, self.attn.qkv.bias = _tn(3 * dim, dim, gen=gen), z(3 * dim)
        self.attn.proj.weight, self.attn.proj.bias = _tn(dim, dim, gen=gen), z(dim)
        self.ls1.gamma, self.ls2.gamma = o(dim), o(dim)
        if ffn == "swiglu":  # published SwiGLUFFNFused: hidden = round_up_8(int(4*dim * 2/3)); w12 = [x1 | x2], w3
            hidden = (int(4 * dim * 2 / 3) + 7) // 8 * 8
            del self.mlp.fc1, self.mlp.fc2
            self.mlp.w12, self.mlp.w3 = _Holder(), _Holder()
            self.mlp.w12.weight, self.mlp.w12.bias = _tn(2 * hidden, dim, gen=gen), z(2 * hidden)
            self.mlp.w3.weight, self.mlp.w3.bias = _tn(dim, hidden, gen=gen), z(dim)
        else:
            self.mlp.fc1.weight, self.mlp.fc1.bias = _tn(4 * dim, dim, gen=gen), z(4 * dim)
            self.mlp.fc2.weight, self.mlp.fc2.bias = _tn(dim, 4 * dim, gen=gen), z(dim)
        self._prep = None
        self._prep16 = None
        self.precision = None  # GEMM precision of this frozen block: None (global default) / ops.PREC_F32 / PREC_BF16X3 / PREC_BF16

    def transposed(self, name):
        """cached [K, N] copy of a prepared weight for the NT-only bf16 kernels' dgrad; None in the exact mode.
        The four copies of a block are made together the first time one is asked for, and the host waits for them: the two
        loss chains re-use the same frozen blocks from two HIP streams with no dependency between them, so a copy enqueued by
        one chain must be complete before the other chain can even enqueue a read (once per block and precision change;
        inside a hipGraph capture the copies exist already from the warm-up steps)."""
        prec = ops.GEMM_PRECISION if self.precision is None else self.precision
        if prec != ops.PREC_BF16:  # (exact fp32 and the split-precision kernels read the weight as it lies)
            return None
        p = self.prepared()
        if "t_" + name not in p:
            with torch.no_grad():
                for k in ("wqkv", "wproj", "w1", "w2"):
                    p["t_" + k] = ops.transpose(p[k])
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
        return p["t_" + name]

    def prepared(self):
        """Kernel-ready frozen weights; LayerScale is folded into the projections once (the teacher never trains):
        x + ls * (a W^T + b) == x + a (ls[:,None] * W)^T + ls * b."""
        w = self.attn.qkv.weight
        if self._prep is None or self._prep["wqkv"].data_ptr() != w.data_ptr() or self._prep["_ver"] != w._version:
            with torch.no_grad():
                ls1, ls2 = self.ls1.gamma, self.ls2.gamma
                self._prep = {
                    "n1w": self.norm1.weight, "n1b": self.norm1.bias, "n2w": self.norm2.weight, "n2b": self.norm2.bias,
                    "wqkv": w, "bqkv": self.attn.qkv.bias,
                    "wproj": (self.attn.proj.weight * ls1[:, None]).contiguous(), "bproj": self.attn.proj.bias * ls1,
                    "_ver": w._version, "swiglu": self.ffn == "swiglu",
                }
                fin, fout = (self.mlp.w12, self.mlp.w3) if self.ffn == "swiglu" else (self.mlp.fc1, self.mlp.fc2)
                self._prep.update({"w1": fin.weight, "b1": fin.bias, "w2": (fout.weight * ls2[:, None]).contiguous(),
                                   "b2": fout.bias * ls2})
        return self._prep

    def prepared16(self):
        """bf16 copies of the frozen weights / biases for the autocast forward (made once): torch autocast casts weight AND
        bias of every nn.Linear to the 16-bit type; LayerScale stays a separate fp32 multiply after the Linear's rounding.
        The copies are enqueued on whichever stream asks first and then handed to EVERY stream (teacher stream, main loss chain,
        frequency-loss chain), which share no dependency for them: the host waits once for the casts, exactly as transposed()
        does (once per block and weight change; during hipGraph capture the copies exist already from the warm-up steps)."""
        w = self.attn.qkv.weight
        if self._prep16 is None or self._prep16["_ptr"] != w.data_ptr() or self._prep16["_ver"] != w._version:
            with torch.no_grad():
                b = lambda t: t.detach().to(torch.bfloat16).contiguous()
                fin, fout = (self.mlp.w12, self.mlp.w3) if self.ffn == "swiglu" else (self.mlp.fc1, self.mlp.fc2)
                self._prep16 = {"_ptr": w.data_ptr(), "_ver": w._version, "swiglu": self.ffn == "swiglu",
                                "n1w": self.norm1.weight, "n1b": self.norm1.bias, "n2w": self.norm2.weight, "n2b": self.norm2.bias,
                                "wqkv": b(w), "bqkv": b(self.attn.qkv.bias), "wproj": b(self.attn.proj.weight),
                                "bproj": b(self.attn.proj.bias), "ls1": self.ls1.gamma, "ls2": self.ls2.gamma,
                                "w1": b(fin.weight), "b1": b(fin.bias), "w2": b(fout.weight), "b2": b(fout.bias)}
            if w.is_cuda and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().synchronize()
        return self._prep16

    def forward_autocast(self, x2, B, N):
        """one block of the autocast teacher on the fp32 residual stream x2 [B*N, D] (no autograd: the frozen teacher's own
        forward).  Op policy of torch.autocast: LayerNorm / softmax / residual adds in fp32, every nn.Linear on bf16 operands
        with a bf16 result, GELU / SiLU on the bf16 tensor."""
        p, H = self.prepared16(), self.num_heads
        D = x2.shape[1]
        hd = D // H
        h1 = ops.layernorm_fwd_bf16out(x2, p["n1w"], p["n1b"], 1e-6)
        # q, k, v as bf16 VALUES in an fp32 buffer: the flash kernel multiplies them exactly (bf16 x bf16 products are exact in
        # fp32), rounds logits and probabilities to bf16 where eager attention under autocast does (matmul result / operand;
        # the softmax itself is fp32) and accumulates in fp32; q * hd^-0.5 is exact for the 64-wide heads of every DINOv2 model
        if hd == 64 and FLASH16:   # every DINOv2 model: the bf16-MFMA flash kernel on the qkv Linear's own 16-bit output
            o16 = ops.attention16_flash(ops.gemm_bf16(h1, p["wqkv"], bias=p["bqkv"]), B, H, N, hd, hd ** -0.5)
        else:
            qkv = ops.gemm_bf16(h1, p["wqkv"], epi=ops.EPI_F32, bias=p["bqkv"])
            st = (N * 3 * D, 3 * D)
            o, _ = ops.attention_forward(qkv, qkv[:, D:], qkv[:, 2 * D:], H, hd, hd ** -0.5, B, N, N, st, st, st, round_bf16=True)
            o16 = ops.cast_bf16(o.reshape(B * N, D))
        x1 = ops.gemm_bf16(o16, p["wproj"], epi=ops.EPI_F32_RESID, bias=p["bproj"], colscale=p["ls1"], residual=x2)
        h2 = ops.layernorm_fwd_bf16out(x1, p["n2w"], p["n2b"], 1e-6)
        if p["swiglu"]:
            a = ops.swiglu_bf16(ops.gemm_bf16(h2, p["w1"], epi=ops.EPI_BF16, bias=p["b1"]))
        else:
            a = ops.gemm_bf16(h2, p["w1"], epi=ops.EPI_BF16_GELU, bias=p["b1"])
        return ops.gemm_bf16(a, p["w2"], epi=ops.EPI_F32_RESID, bias=p["b2"], colscale=p["ls2"], residual=x1)

    def _apply(self, fn, *a, **k):
        self._prep = None
        self._prep16 = None
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            if getattr(self, "policy16", False):  # block re-use under the autocast policy (functional16.ViTBlockFrozen16Fn)
                from .. import functional16 as Fn16
                return Fn16.ViTBlockFrozen16Fn.apply(x, self)
            return Fn.ViTBlockFrozenFn.apply(x, self)
        return Fn.vit_block_forward(x.contiguous(), self.prepared(), self.num_heads, save=False, prec=self.precision)[0]


class VisionTransformer(nn.Module):
    def __init__(self, cfg, seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        D = cfg["embed_dim"]
        self.embed_dim, self.num_heads, self.patch_size = D, cfg["num_heads"], PATCH
        self.cls_token = _tn(1, 1, D, gen=gen)
        self.pos_embed = _tn(1, 1 + POS_GRID * POS_GRID, D, gen=gen)
        self.mask_token = nn.Parameter(torch.zeros(1, D), requires_grad=False)  # present in the public checkpoints
        self.patch_embed = _Holder()
        self.patch_embed.proj = _Holder()
        self.patch_embed.proj.weight = _tn(D, 3, PATCH, PATCH, gen=gen)
        self.patch_embed.proj.bias = nn.Parameter(torch.zeros(D), requires_grad=False)
        self.blocks = nn.ModuleList([Block(D, cfg["num_heads"], gen, cfg.get("ffn", "mlp")) for _ in range(cfg["depth"])])
        self.norm = _Holder()
        self.norm.weight = nn.Parameter(torch.ones(D), requires_grad=False)
        self.norm.bias = nn.Parameter(torch.zeros(D), requires_grad=False)
        self._pos_cache = {}
        self._pe16 = None
        self._tgraph = {}
        self.precision = None
        self.autocast = False

    def set_precision(self, precision):
        """GEMM arithmetic of the frozen teacher: None / "f32" (exact fp32 MFMA), "bf16x3" (split precision) or "bf16"
        (operands rounded to bf16, fp32 accumulate - BASELINE configs[4]'s "bf16 teacher on MFMA")."""
        self.autocast = precision == "bf16_autocast"
        self._tgraph = {}
        if self.autocast:
            # the teacher's OWN forward follows torch.autocast(bf16) with bf16 activations in HBM (forward_tokens_autocast); the
            # re-use of its blocks on student tokens (with gradient) keeps the bf16-operand kernels of the "bf16" mode
            precision = "bf16"
        self.precision = ops.PRECISIONS[precision] if (precision is None or isinstance(precision, str)) else precision
        for blk in self.blocks:
            blk.precision = self.precision
        return self

    def _apply(self, fn, *a, **k):
        self._pos_cache = {}
        self._pe16 = None
        self._tgraph = {}
        return super()._apply(fn, *a, **k)

    def _pos_table(self, ph, pw):
        """[1+ph*pw, D] table for this grid (one-time set-up, cached): bicubic resize of the stored 37x37 table with
        the published recipe (scale_factor = (P + 0.1) / 37, antialias off); row 0 additionally carries the CLS token."""
        key = (ph, pw, self.pos_embed.data_ptr(), self.pos_embed._version)
        if key not in self._pos_cache:
            with torch.no_grad():
                pe = self.pos_embed.float()
                patch = pe[:, 1:]
                m = int(math.isqrt(patch.shape[1]))
                if not (ph * pw == patch.shape[1] and ph == pw):
                    grid = patch.reshape(1, m, m, -1).permute(0, 3, 1, 2)
                    grid = torch.nn.functional.interpolate(grid, scale_factor=((ph + 0.1) / m, (pw + 0.1) / m),
                                                           mode="bicubic", antialias=False)
                    assert grid.shape[-2:] == (ph, pw)
                    patch = grid.permute(0, 2, 3, 1).reshape(1, ph * pw, -1)
                table = torch.cat([pe[:, :1] + self.cls_token, patch], dim=1)[0].contiguous()
            self._pos_cache = {key: table}
        return self._pos_cache[key]

    def forward_tokens_autocast(self, x):
        """forward_tokens under the reference's default mixed-precision policy (train.py:263: precision=16 == torch autocast),
        in bf16: patch-embed conv and every Linear on bf16 operands with bf16 results, the token / residual stream, LayerNorm
        and softmax in fp32 (adding the fp32 position table promotes the embedded patches back to fp32)."""
        B, _, Himg, Wimg = x.shape
        ph, pw = Himg // PATCH, Wimg // PATCH
        n, D = ph * pw, self.embed_dim
        table = self._pos_table(ph, pw)
        if self._pe16 is None or self._pe16[0] != self.patch_embed.proj.weight._version:
            kp = (3 * PATCH * PATCH + 7) // 8 * 8  # 588 -> 592: K padded to whole 16-byte bf16 pieces
            w16 = torch.zeros((D, kp), dtype=torch.bfloat16, device=x.device)
            w16[:, :3 * PATCH * PATCH] = self.patch_embed.proj.weight.reshape(D, -1).to(torch.bfloat16)
            self._pe16 = (self.patch_embed.proj.weight._version, w16, self.patch_embed.proj.bias.to(torch.bfloat16), kp)
            if not torch.cuda.is_current_stream_capturing():  # (same reason as Block.prepared16: built once, read from any stream)
                torch.cuda.current_stream().synchronize()
        _, w16, b16, kp = self._pe16
        rows16 = ops.cast_bf16(ops.patchify_nchw(x.contiguous(), PATCH), kp)
        tok = torch.empty((B * (n + 1), D), dtype=torch.float32, device=x.device)
        ops.gemm_bf16(rows16, w16, epi=ops.EPI_F32, bias=b16, residual=table[1:], res_mod=n, row_map=(n, n + 1, 1, B * (n + 1)),
                      out=tok)
        ops.set_rows(tok, table[0], B, (n + 1) * D)
        t = tok
        for blk in self.blocks:
            t = blk.forward_autocast(t, B, n + 1)
        out, _, _ = ops.layernorm_fwd(t, self.norm.weight, self.norm.bias, 1e-6, need_stats=False, in_rows=(n, n + 1, 1, B * n))
        return out.view(B, n, D)

    def _forward_tokens_autocast_graph(self, x):
        """forward_tokens_autocast replayed as ONE hipGraph launch.  The frozen teacher's forward is a fixed single-stream sequence of
        ~90 launches with no autograd state, so after two eager calls on a given input shape (which build every cached weight copy) it
        is captured once - static input and output buffers, private memory pool - and each later call is a copy into the input buffer
        + one graph launch: ~2 ms less host enqueue per step, which is what bounds the mixed-precision step on most boxes (NOTES
        9.13).  DKD_TEACHER_GRAPH=0 keeps the eager launches; so does any call made while a surrounding capture is in progress."""
        # (weights loaded or edited after the capture: the graph reads the OLD bf16 copies - start over)
        stamp = (self.patch_embed.proj.weight._version,) + tuple(b.attn.qkv.weight._version for b in self.blocks)
        key = (tuple(x.shape), x.device.index)
        ent = self._tgraph.get(key)
        if ent is None or ent["stamp"] != stamp:
            self._tgraph.pop(key, None)
            while len(self._tgraph) >= TEACHER_GRAPH_SHAPES:  # (each entry owns a private-pool graph: last / partial batches, the
                self._tgraph.pop(next(iter(self._tgraph)))    #  validation batch size ... the oldest shape goes first)
            ent = self._tgraph[key] = {"calls": 0, "stamp": stamp}
        else:
            self._tgraph[key] = self._tgraph.pop(key)  # most recently used last
        if "graph" not in ent:
            ent["calls"] += 1
            if ent["calls"] <= 2:
                return self.forward_tokens_autocast(x)
            cur = torch.cuda.current_stream()
            static_in = torch.empty_like(x)
            static_in.copy_(x)
            cap = torch.cuda.Stream(device=x.device)
            cap.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                # (thread_local: a data-loader pin-memory thread or a communication thread may call the runtime meanwhile)
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    static_out = self.forward_tokens_autocast(static_in)
            cur.wait_stream(cap)
            ent.update(graph=g, inp=static_in, out=static_out)
        ent["inp"].copy_(x)
        ent["graph"].replay()
        # a COPY of the static output: the next replay of this shape overwrites that buffer without any autograd version bump, and a
        # caller may keep teacher features across calls (validation collection, two forwards before one backward, KDLossFn's saved
        # tensors).  One device-to-device copy of [B, P*P, D] next to ~90 launches.
        return ent["out"].clone()

    def forward_tokens(self, x):
        """-> final-norm patch tokens [B, P*P, D], dense (CLS dropped by the norm kernel's row gather)."""
        if self.autocast and not (torch.is_grad_enabled() and x.requires_grad):
            if TEACHER_GRAPH and x.is_cuda and not torch.cuda.is_current_stream_capturing():
                return self._forward_tokens_autocast_graph(x)
            return self.forward_tokens_autocast(x)
        B, _, Himg, Wimg = x.shape
        ph, pw = Himg // PATCH, Wimg // PATCH
        n, D = ph * pw, self.embed_dim
        table = self._pos_table(ph, pw)
        rows = ops.patchify_nchw(x.contiguous(), PATCH)
        tok = torch.empty((B * (n + 1), D), dtype=torch.float32, device=x.device)
        ops.gemm(rows, self.patch_embed.proj.weight.reshape(D, -1), bias=self.patch_embed.proj.bias, out=tok,
                 residual=table[1:], res_mod=n, row_map=(n, n + 1, 1, B * (n + 1)), precision=self.precision)
        ops.set_rows(tok, table[0], B, (n + 1) * D)
        t = tok.view(B, n + 1, D)
        for blk in self.blocks:
            t = blk(t)
        out, _, _ = ops.layernorm_fwd(t.reshape(B * (n + 1), D), self.norm.weight, self.norm.bias, 1e-6,
                                      need_stats=False, in_rows=(n, n + 1, 1, B * n))
        return out.view(B, n, D)


class DINOv2ViT(nn.Module):
    def __init__(self, model_name="dinov2_vitg14", seed=0, precision=None):
        super().__init__()
        self.model = VisionTransformer(VIT_CONFIGS[model_name], seed=seed).set_precision(precision)
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False
        self.H = None
        self.W = None

    def forward(self, x):
        with torch.no_grad():
            tok = self.model.forward_tokens(x)
        if self.H is None:  # cached from the first batch like the reference (dinov2.py:34-36)
            self.H, self.W = x.shape[2] // PATCH, x.shape[3] // PATCH
        self.B, _, self.D = tok.shape
        return {"feature_map": tok.reshape(self.B, self.H, self.W, self.D).permute(0, 3, 1, 2)}
