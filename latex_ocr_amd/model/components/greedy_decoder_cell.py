"""GreedyDecoderCell of the reference (model/components/greedy_decoder_cell.py:9-70): initialize / step / finalize."""
import collections

import numpy as np


class DecoderOutput(collections.namedtuple("DecoderOutput", ("logits", "ids"))):       # greedy_decoder_cell.py:5-6
    pass


class GreedyDecoderCell(object):
    def __init__(self, attention_cell, end_token):
        self._attention_cell = attention_cell
        self._batch_size = attention_cell.batch_size
        self._end_token = int(end_token)

    @property
    def output_dtype(self):
        return DecoderOutput(logits="float32", ids="int32")

    @property
    def final_output_dtype(self):
        return self.output_dtype

    def initial_state(self):
        return self._attention_cell.initial_state()

    def initial_inputs(self):
        """greedy_decoder_cell.py:40-43: the start token for every row (row V of the device-side token table)."""
        return "start_token"

    def initialize(self, maximum_iterations=151):
        state = self._attention_cell.begin(1, max_steps=int(maximum_iterations) + 1)
        self._fed = None
        return state, self.initial_inputs(), np.zeros(self._batch_size, dtype=bool)

    def step(self, time, state, embedding, finished):
        """greedy_decoder_cell.py:53-66: logits of the attention cell, ids = int32(argmax), next input = table[ids],
        finished |= (ids == END).  `state` must be the cell's current state tokens (what initialize() / the previous step() returned) or
        an AttentionState of host arrays (uploaded first); `embedding` the ids this cell handed out last (they are still on the device),
        "start_token", or other ids (uploaded).  Finished rows keep stepping, as in the reference."""
        cell = self._attention_cell
        kind = cell.check_state(state)
        own_ids = embedding is self._fed or (isinstance(embedding, str) and embedding == "start_token" and int(time) == 0)
        if kind == "host" or not own_ids:
            # a state / input of the caller's choosing: uploaded into the slot step `time` reads (at time 0 the input is the start token,
            # whatever `embedding` says: greedy_decoder_cell.py:40-43)
            if kind == "host":
                cell._time = int(time) - 1
                cell.load_state(state, embedding)
            elif not isinstance(embedding, str):
                cell.load_state(cell.read_state(), embedding)
        ids, _, fin, logits = cell._engine.decode_step(int(time), self._end_token)
        self._fed = ids
        return DecoderOutput(logits, ids), cell.advance(int(time)), ids, fin

    def finalize(self, final_outputs, final_state):
        return final_outputs
