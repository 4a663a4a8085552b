"""Which co-runner makes the SLP-vectorised resize_fwd_kernel return different bits?  (second half: the standalone
packed-instruction reproducer tests/diag/slp_repro.hip does NOT reproduce beside a pure bf16-MFMA loop, so the round-2 finding is
bisected over the kernels the round-2 co-runner - the ViT-g teacher forward in precision="bf16" - is made of.)

Run ON THE GPU BOX:    python tests/diag/slp_bisect.py
conv.hip is compiled twice into stand-alone shared objects, with the shipped flags (-fno-slp-vectorize) and without that flag (the
round-2 build); dkd_resize_bilinear_fwd_f32 of each is called through ctypes on the current stream while one co-runner class at a time
runs on a side stream, and every output word is compared with the output of the same call made alone."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dinov2_distillation_amd import build, ops  # noqa: E402
from dinov2_distillation_amd.models import DINOv2ViT  # noqa: E402


def conv_lib(slp):
    out = "/tmp/libconv_%s.so" % ("slp" if slp else "noslp")
    flags = [f for f in build.FLAGS if slp is False or f != "-fno-slp-vectorize"]
    subprocess.run([build._hipcc(), *flags, "-shared", os.path.join(build.CSRC, "conv.hip"), "-o", out], check=True, capture_output=True)
    lib = ctypes.CDLL(out)
    lib.dkd_resize_bilinear_fwd_f32.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    return lib


def main():
    dev = "cuda"
    torch.manual_seed(0)
    libs = {"slp": conv_lib(True), "noslp": conv_lib(False)}
    B = 32
    x4, x3 = torch.randn(B, 7, 7, 1024, device=dev), torch.randn(B, 14, 14, 512, device=dev)
    teacher = DINOv2ViT("dinov2_vitg14", seed=0, precision="bf16").cuda()
    xb = torch.randn(32, 3, 224, 224, device=dev)
    with torch.no_grad():
        teacher(xb)
    a32, w32 = torch.randn(8224, 1536, device=dev), torch.randn(4096, 1536, device=dev) * 0.02
    a16, w16 = a32.bfloat16(), w32.bfloat16()
    qkv = torch.randn(B * 257, 3 * 1536, device=dev)
    st = (257 * 3 * 1536, 3 * 1536)
    h = torch.randn(8224, 4096, device=dev)
    g1, b1 = torch.randn(1536, device=dev), torch.randn(1536, device=dev)

    def rep(f, n):
        def run():
            for _ in range(n):
                f()
        return run

    corunners = {
        "nothing": lambda: None,
        "ViT-g teacher forward, precision=bf16 (the round-2 co-runner)": lambda: teacher(xb),
        "its GEMMs only: dkd_gemm_f32 PREC_BF16 (register-staged bf16 MFMA)": rep(lambda: ops.gemm(a32, w32, precision=ops.PREC_BF16), 12),
        "exact fp32 MFMA GEMMs": rep(lambda: ops.gemm(a32, w32, precision=ops.PREC_F32), 4),
        "bf16x3 ring GEMMs (csrc/gemm_x3.hip)": rep(lambda: ops.gemm(a32, w32, precision=ops.PREC_BF16X3), 6),
        "bf16 ring GEMMs (bf16 operands in HBM)": rep(lambda: ops.gemm_bf16(a16, w16), 20),
        "fp32 attention kernels": rep(lambda: ops.attention_forward(qkv, qkv[:, 1536:], qkv[:, 3072:], 24, 64, 0.125, B, 257, 257, st, st, st), 12),
        "VALU / HBM kernels (gelu, layernorm)": rep(lambda: (ops.gelu_fwd(h), ops.layernorm_fwd(a32, g1, b1, 1e-6)), 20),
    }
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    print("%-72s %-6s %s" % ("co-runner", "build", "resize 7->16 / 14->16: differing elements in 6 repetitions"))
    for cname, co in corunners.items():
        for lname, lib in libs.items():
            res = []
            for x in (x4, x3):
                Bx, Hi, Wi, C = x.shape
                out = torch.empty(Bx, 16, 16, C, device=dev)

                def call(o):
                    rc = lib.dkd_resize_bilinear_fwd_f32(x.data_ptr(), o.data_ptr(), Bx, Hi, Wi, 16, 16, C, main_s.cuda_stream)
                    assert rc == 0, rc
                call(out)
                torch.cuda.synchronize()
                ref = out.clone()
                worst = 0
                for _ in range(6):
                    o = torch.empty_like(out)
                    side.wait_stream(main_s)
                    with torch.cuda.stream(side), torch.no_grad():
                        co()
                    call(o)
                    torch.cuda.synchronize()
                    worst = max(worst, int((o != ref).sum()))
                res.append(worst)
            print("%-72s %-6s %6d / %6d" % (cname[:72], lname, res[0], res[1]), flush=True)


if __name__ == "__main__":
    main()
