"""BeamSearchDecoderCell of the reference (model/components/beam_search_decoder_cell.py:43-250): initialize / step /
finalize, incl. the reference's `finalize`, whose loop carries the identity `parents` (quirk C-1 of SURVEY.md), and the
optional true back-trace."""
import collections

import numpy as np

from .greedy_decoder_cell import DecoderOutput


class BeamSearchDecoderCellState(collections.namedtuple("BeamSearchDecoderCellState", ("cell_state", "log_probs"))):
    pass


class BeamSearchDecoderOutput(collections.namedtuple("BeamSearchDecoderOutput", ("logits", "ids", "parents"))):
    pass


class BeamSearchDecoderCell(object):
    def __init__(self, attention_cell, end_token, beam_size=5, div_gamma=1, div_prob=0, div_seed=0, backtrace=False):
        self._cell = attention_cell
        self._batch_size = attention_cell.batch_size
        self._beam_size = int(beam_size)
        self._end_token = int(end_token)
        self._div_gamma, self._div_prob, self._div_seed = float(div_gamma), float(div_prob), int(div_seed)
        self._backtrace = bool(backtrace)

    @property
    def output_dtype(self):
        return BeamSearchDecoderOutput(logits="float32", ids="int32", parents="int32")

    @property
    def final_output_dtype(self):
        return DecoderOutput(logits="float32", ids="int32")

    def initial_state(self):
        """beam_search_decoder_cell.py:98-109: the cell state tiled over the beam, zero log-probs."""
        return BeamSearchDecoderCellState(self._cell.initial_state(), "beam_lp")

    def initial_inputs(self):
        return "start_token"

    def initialize(self, maximum_iterations=151):
        state = self._cell.begin(self._beam_size, max_steps=int(maximum_iterations) + 1, div_gamma=self._div_gamma,
                                 div_prob=self._div_prob, div_seed=self._div_seed)
        return (BeamSearchDecoderCellState(state, "beam_lp"), self.initial_inputs(),
                np.zeros((self._batch_size, self._beam_size), dtype=bool))

    def step(self, time, state, embedding, finished):
        """beam_search_decoder_cell.py:123-187: cell step on batch x beam rows, log-softmax, finished-beam masking, (diversity
        penalty,) top-k over beam x vocabulary (beam 0 only at time 0), ids / parents, state and flags gathered by parents.
        The running log-probs and finished flags stay on the device, so `state.cell_state` must be the cell's CURRENT tokens
        (a stale or foreign token raises; a beam search cannot be re-entered from host arrays without its log-probs)."""
        if self._cell.check_state(state.cell_state) != "current":
            raise ValueError("BeamSearchDecoderCell.step: the beam state (log-probs, finished flags) lives on the device; pass the state "
                             "the previous step returned")
        ids, par, fin, logits = self._cell._engine.decode_step(int(time), self._end_token)
        new_state = BeamSearchDecoderCellState(self._cell.advance(int(time)), "beam_lp")
        return BeamSearchDecoderOutput(logits, ids, par), new_state, ids, fin

    def finalize(self, final_outputs, final_state):
        """final_outputs: time-major stacks [T, B, k(, V)].  Reference-faithful (:189-250): the loop body gathers by the INITIAL
        identity parents at every step, so hypothesis i is ids[:, :, i] unchanged.  backtrace=True follows the parents."""
        ids, par = final_outputs.ids, final_outputs.parents
        if not self._backtrace:
            return DecoderOutput(logits=final_outputs.logits, ids=ids)
        T, B, k = ids.shape
        out = np.empty_like(ids)
        cur = np.tile(np.arange(k)[None, :], (B, 1))
        rows = np.arange(B)[:, None]
        for t in range(T - 1, -1, -1):
            out[t] = ids[t][rows, cur]
            cur = par[t][rows, cur]
        return DecoderOutput(logits=final_outputs.logits, ids=out)
