"""Every launcher of include/dkd_hip.h through plain ctypes, no GPU needed (run by tests/test_abi_cpu.py, once against the shipped
library and once against the AddressSanitizer build of the host side):
  pass 1 - NULL pointers, zero sizes: a negative DKD_ERR_* must come back, nothing may be launched, nothing may crash;
  pass 2 - plausible sizes with fake (never dereferenced on the host) 16-byte-aligned device addresses: the argument checks, launch
           planning and workspace arithmetic run to the launch itself, which fails with a hipError_t on a box without a GPU (positive
           status) - or the call is refused (negative).  Either way the host code has executed under the sanitizer.
usage: python tests/abi_sweep.py <path to libdkd_hip*.so>"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dinov2_distillation_amd import _lib  # noqa: E402  (header parser only: no torch import happens here)

# Pass 2 hands FAKE device addresses to real launchers: with a GPU in reach those launches would run and fault.  The sweep must never see
# one: the caller hides the devices (tests/test_abi_cpu.py sets HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES in the child's environment); this
# is the second lock - hide them here too, before the HIP runtime is loaded, and skip pass 2 if a device is visible all the same.
os.environ["HIP_VISIBLE_DEVICES"] = "-1"
os.environ["ROCR_VISIBLE_DEVICES"] = ""
lib = ctypes.CDLL(sys.argv[1])


def _visible_devices():
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        rc = hip.hipGetDeviceCount(ctypes.byref(n))
        return n.value if rc == 0 else 0
    return 0


GPU_IN_REACH = _visible_devices() > 0
FAKE = 0x7F0000001000   # aligned, unmapped: the host side must never read through a device pointer
INTS = {"B": 2, "H": 14, "W": 14, "C": 64, "O": 64, "k": 3, "stride": 1, "pad": 1, "M": 256, "N": 128, "K": 128, "rows": 256, "cols": 128,
        "groups": 2, "Hi": 14, "Wi": 14, "Ho": 16, "Wo": 16, "hd": 32, "Nq": 64, "Nk": 64, "heads": 2, "lda": 128, "ldb": 128, "ldc": 128,
        "ldr": 128, "ldq": 64, "ldkv": 64, "ldo": 64, "lddo": 64, "lddq": 64, "lddkv": 64, "n": 1024, "R": 16, "batch": 1}


def value(name, ty, real):
    if ty is ctypes.c_void_p:
        return ctypes.c_void_p(FAKE) if real else None
    if ty is ctypes.c_float:
        return ctypes.c_float(1.0 if real else 0.0)
    return ty(INTS.get(name, 8) if real else 0)


n_calls = 0
for real in (False, True):
    if real and GPU_IN_REACH:
        print("SKIP pass 2: a GPU is visible to this process (fake device addresses must never reach one)")
        break
    for name, (restype, args) in _lib._FUNC_DEFS.items():
        launcher = any(a == "stream" for a, _ in args)
        if not launcher and not real:
            continue
        fn = getattr(lib, name)
        fn.restype = ctypes.c_long if restype is ctypes.c_long else ctypes.c_int
        call, keep = [], []
        for an, ty in args:
            if isinstance(ty, str):
                S = _lib.STRUCTS[ty]()
                if real:
                    for fname, fty in _lib._STRUCT_DEFS[ty]:
                        setattr(S, fname, value(fname, fty, True).value)
                keep.append(S)
                call.append(ctypes.byref(S))
            elif an == "stream":
                call.append(None)
            else:
                call.append(value(an, ty, real))
        print("CALL", "real" if real else "null", name, flush=True)
        rc = fn(*call)
        n_calls += 1
        if launcher and not real and rc >= 0:
            print(f"FAIL {name}: NULL arguments were not refused (status {rc})")
            sys.exit(3)
print(f"OK {n_calls} calls")
