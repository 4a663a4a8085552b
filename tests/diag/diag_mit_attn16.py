"""diagnostic: the packed-kv flash policy attention (ops.attention16_cross) against (1) the same ops on the CPU in bf16 (what autocast
runs for MiT's attention core) and (2) the materialised path, for Nq != Nk and non-power-of-two scales"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dinov2_distillation_amd import ops

rel = lambda a, b: ((a.float().cpu() - b.float().cpu()).norm() / (b.float().cpu().norm() + 1e-30)).item()
for (B, H, Nq, Nk, hd) in [(4, 2, 256, 16, 32), (4, 5, 64, 16, 32), (4, 8, 16, 16, 32), (2, 1, 1024, 16, 32), (2, 4, 64, 64, 40), (4, 2, 256, 256, 32)]:
    D = H * hd
    g = torch.Generator().manual_seed(Nq + Nk + hd)
    q = torch.randn(B, Nq, D, generator=g).bfloat16()
    kv = torch.randn(B, Nk, 2 * D, generator=g).bfloat16()
    dO = torch.randn(B, Nq, D, generator=g).bfloat16()
    scale = hd ** -0.5
    qc, kvc = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    qh = qc.reshape(B, Nq, H, hd).permute(0, 2, 1, 3)
    kvh = kvc.reshape(B, Nk, 2, H, hd).permute(2, 0, 3, 1, 4)
    attn = ((qh @ kvh[0].transpose(-2, -1)) * scale)
    P = attn.float().softmax(dim=-1)
    o = (P.bfloat16() @ kvh[1]).transpose(1, 2).reshape(B, Nq, D)
    o.backward(dO)
    qg, kvg = q.cuda(), kv.cuda()
    O, st = ops.attention16_cross(qg.view(B * Nq, D), kvg, B, H, Nq, Nk, hd, scale)
    dq, dkv = ops.attention16_cross(qg.view(B * Nq, D), kvg, B, H, Nq, Nk, hd, scale, dO=dO.cuda().view(B * Nq, D), stats=st)
    k2, v2 = kvg[..., :D].contiguous().view(B * Nk, D), kvg[..., D:].contiguous().view(B * Nk, D)
    ops.ATTN16_FLASH = False
    Om, S, Pm = ops.attention16_forward(qg.view(B * Nq, D), k2, v2, B, H, Nq, Nk, hd, scale)
    dqm, dkm, dvm = ops.attention16_backward(dO.cuda().view(B * Nq, D), qg.view(B * Nq, D), k2, v2, S, Pm, B, H, Nq, Nk, hd, scale)
    ops.ATTN16_FLASH = True
    torch.cuda.synchronize()
    dkvm = torch.cat([dkm.view(B, Nk, D), dvm.view(B, Nk, D)], dim=2)
    print(f"B{B} H{H} Nq{Nq} Nk{Nk} hd{hd}: flash vs cpu  O {rel(O.view(B, Nq, D), o):.2e} dq {rel(dq.view(B, Nq, D), qc.grad):.2e} "
          f"dk {rel(dkv.view(B, Nk, 2 * D)[..., :D], kvc.grad[..., :D]):.2e} dv {rel(dkv.view(B, Nk, 2 * D)[..., D:], kvc.grad[..., D:]):.2e} | "
          f"materialised vs cpu O {rel(Om.view(B, Nq, D), o):.2e} dq {rel(dqm.view(B, Nq, D), qc.grad):.2e} dkv {rel(dkvm, kvc.grad):.2e}")
