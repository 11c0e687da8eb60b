"""dynamic_decode of the reference (model/components/dynamic_decode.py:17-74): the host loop over a decoder cell."""
import numpy as np


def transpose_batch_time(t):
    """dynamic_decode.py:7-15."""
    a = np.asarray(t)
    if a.ndim < 2 or a.ndim > 4:
        raise NotImplementedError
    return np.swapaxes(a, 0, 1)


def _map(fn, *structs):
    s0 = structs[0]
    if hasattr(s0, "_fields"):
        return type(s0)(*[_map(fn, *[getattr(s, f) for s in structs]) for f in s0._fields])
    return fn(*structs)


def dynamic_decode(decoder_cell, maximum_iterations):
    """-> (final_outputs batch-major, final_state).  while not all(finished): step; write outputs[time];
    finished |= (time >= maximum_iterations)  (:38-51), so at most maximum_iterations + 1 steps; finalize; transpose (:70-73).
    One host synchronisation per step (the protocol hands `finished` back to the host loop); lxo_greedy_decode /
    lxo_beam_decode run the same loop with chunked polling and are what Img2SeqModel uses."""
    maximum_iterations = int(maximum_iterations)
    state, inputs, finished = decoder_cell.initialize(maximum_iterations)
    outputs = _map(lambda d: [], decoder_cell.output_dtype)
    time = 0
    while not bool(np.all(finished)):
        new_output, state, inputs, new_finished = decoder_cell.step(time, state, inputs, finished)
        _map(lambda ta, out: ta.append(None if out is None else np.array(out)), outputs, new_output)
        finished = np.logical_or(time >= maximum_iterations, new_finished)
        time += 1
    final_outputs = _map(lambda ta: np.stack(ta, axis=0), outputs)
    final_outputs = decoder_cell.finalize(final_outputs, state)
    final_outputs = _map(transpose_batch_time, final_outputs)
    return final_outputs, state
