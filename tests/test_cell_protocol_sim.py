"""CPU (hipsim): the decoder-cell protocol of latex_ocr_amd/model/components -- AttentionCell.step(embedding, state) from the cell's own
state tokens and from a host state of the caller's choosing (attention_cell.py:58-89), GreedyDecoderCell / dynamic_decode over it
(greedy_decoder_cell.py:46-66, dynamic_decode.py:17-74) -- driven through the C ABI of the host build of the shipped sources.  The same
assertions run on MI355X in tests/test_gpu_refgold.py against the reference-code fixtures."""
import ctypes

import numpy as np
import pytest

from latex_ocr_amd import _abi, synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.components import AttentionCell, GreedyDecoderCell, BeamSearchDecoderCell, dynamic_decode
from latex_ocr_amd.model.utils.image import pad_batch_images
from simlib import SIM_SO, build_sim

SMALL = dict(C=128, E=128, U=128, O=128, D=16)
V = 11


@pytest.fixture(scope="module")
def setup():
    build_sim()
    eng = Engine(V, dims=SMALL, dtype="f32", device="cpu", seed=3, lib=_abi.bind(ctypes.CDLL(SIM_SO)))
    imgs, _ = synthetic.make_set(2, 32, 48, V, 2, 4, seed=5)
    cfg = {"dim_e": SMALL["E"], "dim_o": SMALL["O"], "num_units": SMALL["U"], "dim_embeddings": SMALL["D"]}
    return eng, pad_batch_images(imgs), cfg


def test_attention_cell_step_tokens_and_host_states(setup):
    eng, img, cfg = setup
    cell = AttentionCell(eng, img, cfg, V)
    g = GreedyDecoderCell(cell, V - 1)
    state, inputs, fin = g.initialize(6)
    first, outs = state, []
    for time in range(4):
        out, state, inputs, fin = g.step(time, state, inputs, fin)
        outs.append(out)
        if time == 1:
            saved, saved_ids, saved_tokens = cell.read_state(state), inputs.copy(), state
    ref = eng.greedy_decode(img, V - 1, max_iter=6)              # the device-side loop
    assert np.array_equal(np.stack([o.ids for o in outs], 1), ref[:, :4])
    with pytest.raises(ValueError, match="stale"):
        cell.step(saved_ids, saved_tokens)
    with pytest.raises(ValueError, match="stale"):
        g.step(4, first, inputs, fin)
    with pytest.raises(ValueError, match="foreign"):
        cell.step("start_token", AttentionCell(eng, img, cfg, V).initial_state())
    with pytest.raises(TypeError):
        cell.step(np.zeros((2, SMALL["D"]), np.float32), state)
    cell.begin(1, max_steps=7)
    logits, st = cell.step(saved_ids, saved)                     # the cell alone, from the state saved after step 1
    assert np.abs(logits - outs[2].logits).max() <= 1e-5 and np.array_equal(logits.argmax(1), outs[2].ids)
    logits3, _ = cell.step(outs[2].ids, st)
    assert np.abs(logits3 - outs[3].logits).max() <= 1e-5
    s0, i0, f0 = g.initialize(6)                                 # the greedy cell re-entered at time 2 from the host state
    out2, s2, i2, f2 = g.step(2, saved, saved_ids, f0)
    assert np.array_equal(out2.ids, ref[:, 2])
    out3, _, _, _ = g.step(3, s2, i2, f2)
    assert np.array_equal(out3.ids, ref[:, 3])
    l0, _ = cell.step("start_token", cell.begin(1, max_steps=7))
    assert np.abs(l0 - outs[0].logits).max() <= 1e-5


def test_dynamic_decode_over_the_cells_equals_the_device_loops(setup):
    eng, img, cfg = setup
    cell = AttentionCell(eng, img, cfg, V)
    out, _ = dynamic_decode(GreedyDecoderCell(cell, V - 1), 6)
    assert np.array_equal(out.ids, eng.greedy_decode(img, V - 1, max_iter=6))
    bout, _ = dynamic_decode(BeamSearchDecoderCell(cell, V - 1, beam_size=2), 6)
    assert np.array_equal(bout.ids, eng.beam_decode(img, V - 1, 2, max_iter=6))
    bc = BeamSearchDecoderCell(cell, V - 1, beam_size=2)
    st, inp, fin = bc.initialize(6)
    o, st1, inp, fin = bc.step(0, st, inp, fin)
    with pytest.raises(ValueError):
        bc.step(1, st, inp, fin)                                 # the state of time -1: stale
    bc.step(1, st1, inp, fin)
