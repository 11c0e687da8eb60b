"""Test helper: loads tests/hipsim/build/liblxo_sim.so (the product's HIP sources
compiled for the host against the hipsim SIMT interpreter) and offers numpy
<-> pointer glue.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_SO = os.path.join(ROOT, "tests", "hipsim", "build", "liblxo_sim.so")
_lib = None


def build_sim():
    subprocess.check_call(["make", "-s", "-j8", "sim"], cwd=os.path.join(ROOT, "latex_ocr_amd", "csrc"))


def sim():
    global _lib
    if _lib is None:
        build_sim()
        _lib = ctypes.CDLL(SIM_SO)
        _lib.lxo_last_error.restype = ctypes.c_char_p
    return _lib


def ptr(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def f32_to_bf16(a):
    """float32 ndarray -> uint16 bf16 bits, round-to-nearest-even."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r.reshape(a.shape)


def bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32).reshape(b.shape)


def bf16_round(a):
    return bf16_to_f32(f32_to_bf16(a))
