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
        cell = self._attention_cell
        cell._engine.decode_begin(cell._img, 1, max_steps=int(maximum_iterations) + 1)
        return self.initial_state(), self.initial_inputs(), np.zeros(self._batch_size, dtype=bool)

    def step(self, time, state, embedding, finished):
        """greedy_decoder_cell.py:53-66: logits of the attention cell, ids = int32(argmax), next input = table[ids],
        finished |= (ids == END).  `embedding` is the ids fed back (kept on the device); finished rows keep stepping."""
        ids, _, fin, logits = self._attention_cell._engine.decode_step(int(time), self._end_token)
        return DecoderOutput(logits, ids), self._attention_cell.initial_state(int(time)), ids, fin

    def finalize(self, final_outputs, final_state):
        return final_outputs
