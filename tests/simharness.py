"""TEST INFRASTRUCTURE: drives the C ABI of the hipsim build with NumPy buffers."""
import ctypes

import numpy as np

from latex_ocr_amd import _abi
from latex_ocr_amd.model import params as PP
from simlib import SIM_SO, build_sim, ptr, f32_to_bf16, bf16_to_f32

_L = None


def lib():
    global _L
    if _L is None:
        build_sim()
        _L = _abi.bind(ctypes.CDLL(SIM_SO))
    return _L


class Sim(object):
    """One shape + buffers (params, packed weights, workspace, grads) on the host."""

    def __init__(self, B, H, W, T, V, dtype=0, dims=None, beam=1, max_steps=0, seed=0, params=None):
        d = dict(PP.DEFAULT_DIMS, **(dims or {}))
        self.L = lib()
        self.shape = _abi.LxoShape(B, H, W, T, V, d["C"], d["E"], d["U"], d["O"], d["D"], dtype, beam, max_steps)
        self.shape.encoder_cnn = 1 if d.get("cnn") else 0
        self.shape.no_positional = 0 if d.get("positional", True) else 1
        self.shape.encoder_rnn = 1 if d.get("row_bilstm") else 0
        self.dims, self.V, self.dtype = d, V, dtype
        self.specs = PP.param_specs(V, d)
        self.P = params if params is not None else PP.init_params(V, seed, d)
        n = self.L.lxo_param_total(ctypes.byref(self.shape))
        assert n == PP.n_params(V, d), (n, PP.n_params(V, d))
        self.params = np.concatenate([self.P[k].reshape(-1) for k, _, _ in self.specs]).astype(np.float32)
        self.grads = np.zeros(n, np.float32)
        self.wpack = np.zeros(self.L.lxo_wpack_bytes(ctypes.byref(self.shape)) + 256, np.uint8)
        self.ws = np.zeros(self.L.lxo_workspace_bytes(ctypes.byref(self.shape)) + 256, np.uint8)
        self.pack()

    def sref(self):
        return ctypes.byref(self.shape)

    def ck(self, rc, what):
        _abi.check(self.L, rc, what)

    def pack(self):
        self.ck(self.L.lxo_pack_weights(self.sref(), ptr(self.params), ptr(self.wpack), None), "pack")

    def set_params(self, P):
        self.params[:] = np.concatenate([np.asarray(P[k], np.float32).reshape(-1) for k, _, _ in self.specs])
        self.pack()

    def region(self, name, dtype, shape=None):
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        self.ck(self.L.lxo_ws_region(self.sref(), name.encode(), ctypes.byref(off), ctypes.byref(nb)), "region")
        raw = self.ws[off.value:off.value + nb.value]
        if dtype == "ct":
            if self.dtype == 1:
                a = bf16_to_f32(raw.view(np.uint16))
            else:
                a = raw.view(np.float32)
        else:
            a = raw.view(dtype)
        if shape is not None:
            a = a[:int(np.prod(shape))].reshape(shape)
        return a

    def region_ptr(self, name):
        off = ctypes.c_size_t()
        self.ck(self.L.lxo_ws_region(self.sref(), name.encode(), ctypes.byref(off), None), "region")
        return ctypes.c_void_p(self.ws.ctypes.data + off.value)

    def write_region(self, name, arr):
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        self.ck(self.L.lxo_ws_region(self.sref(), name.encode(), ctypes.byref(off), ctypes.byref(nb)), "region")
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        assert raw.size <= nb.value
        self.ws[off.value:off.value + raw.size] = raw

    def grad(self, name):
        off = 0
        for k, shp, _ in self.specs:
            n = int(np.prod(shp))
            if k == name:
                return self.grads[off:off + n].reshape(shp)
            off += n
        raise KeyError(name)
