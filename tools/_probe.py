"""ctypes loader of the tools-only probe library (tools/probe/libsvihmm_probe.so)."""
import ctypes as C
import os
import subprocess
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe")


def probe_fp64(which, device=0):
    so = os.path.join(HERE, "libsvihmm_probe.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", HERE])
    lib = C.CDLL(so)
    lib.probe_fp64.restype = C.c_int
    lib.probe_fp64.argtypes = [C.c_int, C.c_int32, C.POINTER(C.c_double)]
    v = C.c_double()
    if lib.probe_fp64(int(device), int(which), C.byref(v)) != 0:
        raise RuntimeError("probe_fp64(%d) failed" % which)
    return v.value
