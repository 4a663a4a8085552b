"""A few forward attention launches of one shape (for the PMC passes of scripts/pmc_attn.sh): Nq Nk [packed]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
Nq, Nk = int(sys.argv[1]), int(sys.argv[2])
B, H, hd = 64, 6, 64
D = H * hd
q = torch.randn(B, Nq, D, device="cuda")
kv = torch.randn(B, Nk, 2 * D, device="cuda")
for _ in range(4):
    ops.attention_forward(q, kv, kv[..., D:], H, hd, hd ** -0.5, B, Nq, Nk, (Nq * D, D), (Nk * 2 * D, 2 * D), (Nk * 2 * D, 2 * D))
torch.cuda.synchronize()
