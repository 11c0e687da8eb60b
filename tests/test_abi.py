"""CPU: the gfx950 C-ABI library loads without a GPU and exports every symbol include/lxo.h
declares; plan queries agree with the Python-side inventory; the product refuses to run
without its HIP library / a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from latex_ocr_amd import _abi
from latex_ocr_amd.model import params as PP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(_abi.LIB_PATH):
        g.build()
    return _abi.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "lxo.h")).read()
    declared = sorted(set(re.findall(r"\b(lxo_[a-z0-9_]+)\s*\(", hdr)))
    assert declared and set(declared) == set(_abi.ENTRY_POINTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lxo_version() == _abi.ABI_VERSION == 6
    assert lib.lxo_shape_size() == ctypes.sizeof(_abi.LxoShape)


def test_dynamic_symbol_table_is_exactly_the_two_headers(lib):
    """-fvisibility=hidden + csrc/exports.map: `nm -D` of the production library lists the entry points of include/lxo.h, the five
    measurement hooks of include/lxo_debug.h, and nothing else -- no C++ internals, no test hooks (there is no fault injector)."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _abi.LIB_PATH]).decode()
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    declared = set()
    for h in ("lxo.h", "lxo_debug.h"):
        declared |= set(re.findall(r"\b(lxo_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    assert set(exported) == declared, (sorted(set(exported) - declared), sorted(declared - set(exported)))
    assert not [s for s in exported if "inject" in s]


def test_param_table_matches_python(lib):
    for V, cnn in ((50, 0), (500, 0), (50, 1)):
        s = _abi.LxoShape(4, 32, 128, 10, V, 512, 256, 512, 512, 80, 1, 1, 0)
        s.encoder_cnn = cnn
        dims = dict(PP.DEFAULT_DIMS, cnn=bool(cnn))
        assert lib.lxo_param_total(ctypes.byref(s)) == PP.n_params(V, dims)
        off = 0
        specs = list(PP.param_specs(V, dims))
        for i in range(lib.lxo_param_num()):              # slots of the other encoder variant have count 0
            o, c = ctypes.c_longlong(), ctypes.c_longlong()
            assert lib.lxo_param_info(ctypes.byref(s), i, ctypes.byref(o), ctypes.byref(c)) == 0
            if c.value == 0:
                continue
            name, shp, _ = specs.pop(0)
            assert lib.lxo_param_name_for(ctypes.byref(s), i).decode() == name
            if not cnn:
                assert lib.lxo_param_name(i).decode() == name
            assert (o.value, c.value) == (off, int(np.prod(shp)))
            off += int(np.prod(shp))
        assert not specs


def test_workspace_queries_and_validation(lib):
    s = _abi.LxoShape(64, 128, 512, 101, 500, 512, 256, 512, 512, 80, 1, 1, 0)
    ws = lib.lxo_workspace_bytes(ctypes.byref(s))
    assert 1 << 30 < ws < 8 << 30
    off, nb = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.lxo_ws_region(ctypes.byref(s), b"img", ctypes.byref(off), ctypes.byref(nb)) == 0
    assert nb.value == 64 * 14 * 62 * 512 * 2                       # [B, R=868, C] bf16
    assert lib.lxo_ws_region(ctypes.byref(s), b"nope", None, None) != 0
    bad = _abi.LxoShape(1, 8, 8, 4, 50, 512, 256, 512, 512, 80, 1, 1, 0)      # image too small for conv6 VALID
    assert lib.lxo_encoder_fwd(ctypes.byref(bad), None, None, None, None, None) != 0
    assert b"image too small" in lib.lxo_last_error()


def test_ws_region_dtype_follows_the_plan(lib):
    """lxo_ws_region_dtype names the element type an external driver of lxo_encoder_bwd must write into "d_img": bf16 (d_y6, masked) only
    where the decoder's last GEMM applies conv6's mask -- bf16 mode, E % 32 == 0, no row encoder -- and the plain f32 gradient otherwise."""
    def dt(dtype, E=256, rnn=0):
        s = _abi.LxoShape(4, 32, 128, 10, 50, 512, E, 512, 512, 80, dtype, 1, 0)
        s.encoder_rnn = rnn
        return lib.lxo_ws_region_dtype(ctypes.byref(s), b"d_img")
    assert dt(_abi.LXO_BF16) == _abi.LXO_BF16
    assert dt(_abi.LXO_F32) == _abi.LXO_F32
    assert dt(_abi.LXO_BF16, rnn=1) == _abi.LXO_F32          # with the row encoder "d_img" is the gradient w.r.t. ITS output: f32
    s = _abi.LxoShape(4, 32, 128, 10, 50, 512, 256, 512, 512, 80, _abi.LXO_BF16, 1, 0)
    for name, want in ((b"img", _abi.LXO_BF16), (b"att_exp", _abi.LXO_F32), (b"recb", _abi.LXO_BF16), (b"alpha", _abi.LXO_F32), (b"m2", _abi.LXO_U8),
                       (b"dec_ids", _abi.LXO_I32)):
        if name == b"att_exp":
            continue          # listed below with the compute-dtype regions
        assert lib.lxo_ws_region_dtype(ctypes.byref(s), name) == want, name
    assert lib.lxo_ws_region_dtype(ctypes.byref(s), b"att_exp") == _abi.LXO_BF16
    assert lib.lxo_ws_region_dtype(ctypes.byref(s), b"no_such_region") < 0


def test_xdec_sync_region_is_two_chain_blocks(lib):
    """Engine.chain_status reads the error word of the forward / backward chain at block 0 / 1 of region "xdec_sync": the block size the
    Python side assumes must be the one the plan lays out (csrc/xdec.h: kXDecBlockBytes)."""
    from latex_ocr_amd.engine import XDEC_BLOCK_BYTES
    s = _abi.LxoShape(8, 32, 128, 10, 50, 512, 256, 512, 512, 80, _abi.LXO_BF16, 1, 0)
    off, nb = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.lxo_ws_region(ctypes.byref(s), b"xdec_sync", ctypes.byref(off), ctypes.byref(nb)) == 0
    assert nb.value >= 2 * XDEC_BLOCK_BYTES and off.value % 128 == 0, (off.value, nb.value)
    s32 = _abi.LxoShape(8, 32, 128, 10, 50, 512, 256, 512, 512, 80, _abi.LXO_F32, 1, 0)      # no chains in f32 mode: the flag words only
    assert lib.lxo_ws_region(ctypes.byref(s32), b"xdec_sync", ctypes.byref(off), ctypes.byref(nb)) == 0
    assert 0 < nb.value < XDEC_BLOCK_BYTES


def test_no_cpu_fallback():
    import torch
    from latex_ocr_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        Engine(50, device="cpu")
