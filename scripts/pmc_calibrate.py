"""PMC calibration (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): kernels with KNOWN byte counts in OUR access
patterns, so the gfx950 FETCH_SIZE half-count correction can be checked per pattern."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_distillation_amd import ops
n = 128 * 1024 * 1024  # 512 MB per array (> 256 MB Infinity Cache)
a, b = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
ops.axpby(a, b, 1.0, 1.0)                      # reads 2 x 512 MiB, writes 512 MiB, float4 per lane
x = torch.randn(200704, 384, device="cuda")    # A: 308.3 MB, read exactly once (one column tile)
w = torch.randn(128, 384, device="cuda")
ops.gemm(x, w)                                 # MK/NK (k-contiguous float4 loads, 64 B per row per k-tile); writes 102.8 MB
ops.gemm(x, torch.randn(384, 128, device="cuda"), b_layout=1)
xt = torch.randn(384, 200704, device="cuda")   # A in KM layout (m-contiguous float4 loads)
ops.gemm(xt, w, a_layout=1)
torch.cuda.synchronize()
