"""Build libdkd_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

    python -m dinov2_distillation_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under build/ and rebuilt when the source (or a header) is
newer.  The .so lands next to the package (dinov2_distillation_amd/lib/) so it travels with a repo snapshot.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
OBJ = os.path.join(ROOT, "build", "obj")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libdkd_hip.so")
ARCH = "gfx950"
# -fno-slp-vectorize: hipcc's SLP vectoriser packs adjacent scalar fp32 arithmetic into v_pk_*_f32.  Round 2 measured a kernel
# built that way (resize_fwd_kernel) returning DIFFERENT bits for the same input whenever bf16-MFMA waves of another stream
# were co-resident (tests/test_corun_gpu.py; the same source without packed math is immune), packed fp32 beside MFMAs is an
# issue-slot anti-lever anyway (MI355X_MICROARCH.md, cycle constants), and the step time is unchanged without it (44.0 ms).
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + INC, "-I" + CSRC, "-Wall", "-Wno-unused-function",
         "-fno-slp-vectorize"]
# per-file code generation switches (measured, see NOTES.md section 3)
# augment.hip reproduces Pillow's float / double arithmetic operation by operation: no fused multiply-add contraction there
# (HIP's __fmul_rn / __dadd_rn are plain operators to the compiler, so a * b + c would still become one fma)
FILE_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "augment.hip": ["-ffp-contract=off"]}
if os.environ.get("DKD_TUNING") == "1":  # also instantiate the timing-only knock-out kernels (scripts/knockout_*.py)
    FLAGS.append("-DDKD_TUNING")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


# AddressSanitizer build of the HOST side ("ASan host build"): the launchers' argument validation, launch planning and
# workspace arithmetic instrumented; device code is compiled as usual (GPU ASan needs xnack+, not available on this pool).  Lives
# under build/asan (never shipped); tests/test_abi_cpu.py runs every launcher through it.
ASAN_OBJ = os.path.join(ROOT, "build", "asan")
ASAN_LIB = os.path.join(ASAN_OBJ, "libdkd_hip_asan.so")
ASAN_FLAGS = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-Wno-option-ignored"]


def asan_runtime():
    """the clang ASan runtime to LD_PRELOAD into an uninstrumented python"""
    r = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    if os.path.isabs(p) and os.path.exists(p):
        return p
    import glob
    c = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return c[0] if c else None


def _compile(src, force, asan=False):
    obj = os.path.join(ASAN_OBJ if asan else OBJ, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_header()):
        return obj, False
    cmd = [_hipcc(), *FLAGS, *(ASAN_FLAGS if asan else []), *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB) or force:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[dkd build] linked {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"[dkd build] {LIB} up to date")
    return LIB


def build_asan(force=False, verbose=True):
    os.makedirs(ASAN_OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, asan=True), srcs))
    if any(ch for _, ch in res) or not os.path.exists(ASAN_LIB) or force:
        cmd = [_hipcc(), "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", f"--offload-arch={ARCH}", "-o", ASAN_LIB, *[o for o, _ in res]]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[dkd build] linked {ASAN_LIB}")
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan(force="--force" in sys.argv)
    else:
        build(force="--force" in sys.argv)
