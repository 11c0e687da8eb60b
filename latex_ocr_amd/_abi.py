"""ctypes declaration of the C ABI in include/lxo.h (signatures only).

`load()` opens the in-tree gfx950 library `latex_ocr_amd/liblxo.so` and fails
loudly when it is missing: there is no CPU fallback in this package.
"""
import ctypes
import os

c_int, c_ll, c_size, c_float, c_void = ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p
P = ctypes.POINTER

LXO_F32, LXO_BF16 = 0, 1
LXO_GNORM_FLOATS = 2 + 1024   # include/lxo.h: floats lxo_global_norm_scale needs behind scale_out
LIB_PATH = os.environ.get("LXO_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblxo.so")   # the override is a measurement aid (diagnostic builds)


ABI_VERSION = 6          # include/lxo.h LXO_ABI_VERSION
LXO_COMM_ID_BYTES = 128
LXO_I32, LXO_U8 = 2, 3
LXO_XDEC_BLOCK_BYTES, LXO_XDEC_ERR_WORD = 4096 + (384 << 10), 512      # include/lxo.h: the chains' error words in ws region "xdec_sync"


class LxoShape(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("B", "H", "W", "T", "V", "C", "E", "U", "O", "D", "dtype", "beam", "max_steps")] + \
               [("keep_prob", c_float), ("dropout_seed", c_int), ("div_gamma", c_float), ("div_prob", c_float), ("div_seed", c_int),
                ("encoder_cnn", c_int), ("no_positional", c_int), ("step_kernels", c_int), ("encoder_rnn", c_int), ("deterministic", c_int), ("live_B", c_int)]


def bind(lib):
    """Attach argtypes/restypes of every entry point of include/lxo.h."""
    S = P(LxoShape)
    sig = {
        "lxo_last_error": (ctypes.c_char_p, []),
        "lxo_version": (c_int, []),
        "lxo_shape_size": (c_int, []),
        "lxo_ws_region_dtype": (c_int, [S, ctypes.c_char_p]),
        "lxo_gemm_nt": (c_int, [c_int] * 4 + [c_void] * 3 + [c_int] * 6 + [c_void, c_int, c_float, c_int, c_void]),
        "lxo_gemm_tn": (c_int, [c_int] * 3 + [c_void] * 3 + [c_int] * 8 + [c_void]),
        "lxo_conv3x3": (c_int, [c_int, c_void, c_void, c_void, c_void] + [c_int] * 9 + [c_void]),
        "lxo_conv3x3_ex": (c_int, [c_int, c_void, c_void, c_void, c_void] + [c_int] * 9 + [c_void, c_int, c_void, c_void, c_void, c_void]),
        "lxo_conv3x3_wgrad": (c_int, [c_int, c_void, c_void, c_void] + [c_int] * 8 + [c_void]),
        "lxo_gemm_slab": (c_int, [c_int, c_void, c_void, c_void] + [c_int] * 6 + [c_ll, c_void]),
        "lxo_attention_fwd": (c_int, [c_int] + [c_void] * 7 + [c_int] * 6 + [c_void]),
        "lxo_timing_enable": (c_int, [c_int]),
        "lxo_timing_count": (c_int, []),
        "lxo_timing_get": (c_int, [c_int, P(ctypes.c_char_p), P(ctypes.c_char_p), P(ctypes.c_double), P(c_float)]),
        "lxo_param_num": (c_int, []),
        "lxo_param_name": (ctypes.c_char_p, [c_int]),
        "lxo_param_name_for": (ctypes.c_char_p, [S, c_int]),
        "lxo_param_total": (c_ll, [S]),
        "lxo_param_info": (c_int, [S, c_int, P(c_ll), P(c_ll)]),
        "lxo_wpack_bytes": (c_size, [S]),
        "lxo_workspace_bytes": (c_size, [S]),
        "lxo_ws_region": (c_int, [S, ctypes.c_char_p, P(c_size), P(c_size)]),
        "lxo_pack_weights": (c_int, [S, c_void, c_void, c_void]),
        "lxo_encoder_fwd": (c_int, [S, c_void, c_void, c_void, c_void, c_void]),
        "lxo_encoder_bwd": (c_int, [S, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_void]),
        "lxo_encoder_bwd_ready": (c_int, [S, c_void, c_void, c_void, c_void, c_void, c_int, c_int, c_void, c_void]),
        "lxo_train_bwd": (c_int, [S, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
        "lxo_set_encoder_side_stream": (c_int, [c_void]),
        "lxo_set_side_stream": (c_int, [c_void]),
        "lxo_decoder_train_fwd": (c_int, [S, c_void, c_void, c_void, c_void, c_void]),
        "lxo_ce_loss_fwd_bwd": (c_int, [S, c_void, c_void, c_void, c_float, c_void]),
        "lxo_ce_loss_fwd_bwd_dev": (c_int, [S, c_void, c_void, c_void, c_void, c_void]),
        "lxo_decoder_train_bwd_part": (c_int, [S, c_void, c_void, c_void, c_void, c_void, c_int, c_void]),
        "lxo_decoder_train_bwd": (c_int, [S, c_void, c_void, c_void, c_void, c_void, c_void]),
        "lxo_global_norm_scale": (c_int, [c_ll, c_void, c_float, c_void, c_void]),
        "lxo_adam_step": (c_int, [c_ll, c_void, c_void, c_void, c_void, c_float, c_float, c_float, c_float, c_void, c_void]),
        "lxo_optimizer_step": (c_int, [c_int, c_ll, c_void, c_void, c_void, c_float, c_void, c_void]),
        "lxo_greedy_decode": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void, P(c_int), c_void]),
        "lxo_greedy_decode_attn": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void, c_void, P(c_int), c_void]),
        "lxo_decode_begin": (c_int, [S, c_void, c_void, c_void, c_void]),
        "lxo_decode_step": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void, c_void, c_void, P(c_int), c_void]),
        "lxo_beam_decode": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void, c_void, P(c_int), c_void]),
        "lxo_beam_decode_attn": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void, c_void, c_void, P(c_int), c_void]),
        "lxo_chain_guard": (c_int, [S, c_void, c_void, c_void, c_int, c_void, c_void]),
        "lxo_decode_state_get": (c_int, [S, c_void, c_int, c_void, c_void, c_void, c_void]),
        "lxo_decode_state_set": (c_int, [S, c_void, c_int, c_void, c_void, c_void, c_void, c_void]),
        "lxo_decode_cell_step": (c_int, [S, c_void, c_void, c_void, c_int, c_int, c_void]),
        "lxo_comm_unique_id": (c_int, [c_void]),
        "lxo_comm_init": (c_int, [c_void, c_int, c_int, P(c_void)]),
        "lxo_comm_info": (c_int, [c_void, P(c_int), P(c_int)]),
        "lxo_allreduce_bucket": (c_int, [c_void, c_void, c_ll, c_int, c_void, c_void]),
        "lxo_comm_destroy": (c_int, [c_void]),
        "lxo_comm_last_error": (ctypes.c_char_p, []),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    lib._lxo_missing = missing
    if not missing:
        # a library built from another revision of include/lxo.h would read past (or short of) the struct this module passes
        if lib.lxo_version() != ABI_VERSION or lib.lxo_shape_size() != ctypes.sizeof(LxoShape):
            raise RuntimeError("liblxo ABI mismatch: library version %d / lxo_shape %d bytes, binding version %d / %d bytes"
                               % (lib.lxo_version(), lib.lxo_shape_size(), ABI_VERSION, ctypes.sizeof(LxoShape)))
    return lib


ENTRY_POINTS = ["lxo_last_error", "lxo_version", "lxo_shape_size", "lxo_ws_region_dtype", "lxo_timing_enable", "lxo_timing_count", "lxo_timing_get", "lxo_gemm_nt", "lxo_gemm_tn", "lxo_conv3x3", "lxo_conv3x3_ex", "lxo_conv3x3_wgrad", "lxo_gemm_slab", "lxo_attention_fwd", "lxo_param_num", "lxo_param_name", "lxo_param_name_for",
                "lxo_param_total", "lxo_param_info", "lxo_wpack_bytes", "lxo_workspace_bytes", "lxo_ws_region",
                "lxo_pack_weights", "lxo_encoder_fwd", "lxo_encoder_bwd", "lxo_encoder_bwd_ready", "lxo_train_bwd", "lxo_set_side_stream", "lxo_set_encoder_side_stream", "lxo_decoder_train_fwd",
                "lxo_ce_loss_fwd_bwd", "lxo_ce_loss_fwd_bwd_dev", "lxo_decoder_train_bwd", "lxo_decoder_train_bwd_part", "lxo_global_norm_scale", "lxo_adam_step", "lxo_optimizer_step",
                "lxo_greedy_decode", "lxo_greedy_decode_attn", "lxo_beam_decode", "lxo_beam_decode_attn", "lxo_decode_begin", "lxo_decode_step",
                "lxo_chain_guard", "lxo_decode_state_get", "lxo_decode_state_set", "lxo_decode_cell_step",
                "lxo_comm_unique_id", "lxo_comm_init", "lxo_comm_info", "lxo_allreduce_bucket", "lxo_comm_destroy", "lxo_comm_last_error"]

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "latex_ocr_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        try:
            import torch  # noqa: F401  -- first, so that the library shares the HIP runtime PyTorch carries (load order)
        except ImportError:
            pass
        _lib = bind(ctypes.CDLL(LIB_PATH))
        if _lib._lxo_missing:
            raise RuntimeError("liblxo.so lacks entry points: %s" % _lib._lxo_missing)
    return _lib


def check(lib, rc, what=""):
    if rc != 0:
        raise RuntimeError("%s failed: rc=%d %s" % (what, rc, lib.lxo_last_error().decode()))
