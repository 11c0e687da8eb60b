"""CPU: checkpoint containers (npz / safetensors / TF Saver bundle restated in latex_ocr_amd/tf_checkpoint.py) and the
LR-schedule state carried by this framework's checkpoints."""
import os
import struct
import sys

import numpy as np

from latex_ocr_amd import tf_checkpoint as TC
from latex_ocr_amd.model import params as PP
from latex_ocr_amd.model.base import BaseModel
from latex_ocr_amd.model.utils.lr_schedule import LRSchedule

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors of CRC-32C
    assert TC.crc32c(b"123456789") == 0xE3069283
    assert TC.crc32c(bytes(32)) == 0x8A9136AA
    assert TC.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43


def test_bundle_round_trip_and_table_layout(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {"Decoder/embedding_table": rng.randn(7, 5).astype(np.float32), "optimize/beta1_power": np.float32(0.81),
               "Encoder/convolutional_encoder/conv2d/kernel": rng.randn(3, 3, 1, 4).astype(np.float32),
               "global_step": np.int64(12), "flags": np.arange(6, dtype=np.int32).reshape(2, 3)}
    for i in range(40):                                  # > 16 entries: several data blocks behind the index block
        tensors["v/%03d" % i] = rng.randn(i % 3 + 1).astype(np.float32)
    prefix = str(tmp_path / "model.cpkt-3")
    TC.write_bundle(prefix, tensors)
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == TC.MAGIC and len(raw) > 48
    got = TC.read_bundle(prefix, verify_crc=True)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == np.asarray(v).dtype and np.array_equal(got[k], v), k
    idx = TC.read_index(prefix + ".index")
    assert idx["Decoder/embedding_table"]["shape"] == (7, 5) and idx["global_step"]["shape"] == ()
    # corrupting a tensor byte is caught by the per-tensor crc
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[idx["flags"]["offset"]] ^= 0xFF
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    try:
        TC.read_bundle(prefix, names={"flags"}, verify_crc=True)
        assert False
    except IOError:
        pass


def test_convert_between_containers(tmp_path):
    import convert_checkpoint as CC
    V = 13
    P = PP.init_params(V, 1, dict(PP.DEFAULT_DIMS, C=128, E=128, U=128, O=128, D=16))
    arrays = {k: np.asarray(v, np.float32) for k, v in P.items()}
    for k, v in P.items():
        arrays["optimize/" + k + "/Adam"] = np.full_like(v, 0.5, dtype=np.float32)
        arrays["optimize/" + k + "/Adam_1"] = np.full_like(v, 0.25, dtype=np.float32)
    arrays["optimize/adam_t"] = np.int64(7)
    src = str(tmp_path / "model.cpkt-2")
    with open(src, "wb") as f:
        np.savez(f, **arrays)
    CC.main([src, str(tmp_path / "m.safetensors")])
    CC.main([str(tmp_path / "m.safetensors"), str(tmp_path / "tfck"), "--to", "tf"])
    back = BaseModel._open_checkpoint(str(tmp_path / "tfck"))
    for k, v in P.items():
        assert np.array_equal(back[k], np.asarray(v, np.float32))
        assert np.array_equal(back["optimize/" + k + "/Adam_1"], arrays["optimize/" + k + "/Adam_1"])
    from latex_ocr_amd.model.base import adam_steps_from_powers
    # TF convention: beta^(t+1) after t steps
    assert abs(float(back["optimize/beta1_power"]) - 0.9 ** 8) < 1e-7
    assert adam_steps_from_powers(back["optimize/beta1_power"], back["optimize/beta2_power"]) == 7
    w = CC.main([src, str(tmp_path / "w.npz"), "--weights-only"])
    assert set(w) == set(P)


def test_lr_schedule_state_round_trip():
    a = LRSchedule(lr_init=1e-3, lr_min=1e-5, start_decay=4, end_decay=20, lr_warm=1e-4, end_warm=2, decay_rate=0.5, early_stopping=3)
    for i in range(9):
        a.update(batch_no=i)
    a.update(score=-3.0); a.update(score=-4.0)
    b = LRSchedule(lr_init=1e-3, lr_min=1e-5, start_decay=4, end_decay=20, lr_warm=1e-4, end_warm=2, decay_rate=0.5, early_stopping=3)
    b.load_state_dict({k: np.float64(v) for k, v in a.state_dict().items()})
    assert b.lr == a.lr and b._score == a._score and b._n_batch_no_imprv == a._n_batch_no_imprv
    for i in range(9, 14):
        a.update(batch_no=i); b.update(batch_no=i)
    a.update(score=-5.0); b.update(score=-5.0)
    assert b.lr == a.lr and b.stop_training == a.stop_training


def test_adam_steps_from_tf_powers_survive_underflow():
    """A realistically trained reference checkpoint: float32 0.9^(t+1) has underflowed to 0."""
    from latex_ocr_amd.model.base import adam_steps_from_powers as f
    assert f(np.float32(0.9), np.float32(0.999)) == 0                       # freshly initialised optimizer
    assert f(np.float32(0.9 ** 11), np.float32(0.999 ** 11)) == 10
    t = 20000
    assert f(np.float32(0.0), np.float32(0.999 ** (t + 1))) in range(t - 2, t + 3)
    assert f(np.float32(0.9 ** (t + 1)), np.float32(0.999 ** (t + 1))) in range(t - 2, t + 3)
    assert f(np.float32(0.0), np.float32(0.0)) >= 100000                     # both gone: bias corrections ~ 1
    assert f(np.float32(0.0), None) >= 100000
    assert f(np.float32(np.nan), np.float32(np.inf)) >= 100000
