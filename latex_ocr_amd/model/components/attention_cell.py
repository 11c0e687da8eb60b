"""AttentionCell of the reference (model/components/attention_cell.py:11-102) as a handle on the device-side cell.

In the reference the cell owns the graph of one step: LSTMCell, dropout, AttentionMechanism.context, the o / logits
projections (attention_cell.py:58-89).  Here that step is one call of the HIP library (lxo_decode_cell_step: csrc/model_decoder.hip
decode_common_step: the fused LSTM / attention / tanh-o kernels + the logits GEMM) and the state
(AttentionState(cell_state=LSTMStateTuple(c, h), o)) lives on the GPU.  The object hands out STATE TOKENS (DeviceState) that name the
state of one time step; `step(embedding, state)` accepts

* the tokens of the cell's CURRENT state (the fast path: nothing moves), and raises on stale or foreign tokens -- a token names a state
  that may no longer exist on the device, and stepping "from" it would silently step from something else;
* an AttentionState of host arrays (c [N, U], h [N, U], o [N, O] float32): the caller's own state, uploaded with
  lxo_decode_state_set -- so a caller can step from a state of its choosing, as the reference's cell allows;

`read_state(state)` turns the current tokens into such host arrays (lxo_decode_state_get)."""
import collections

import numpy as np

AttentionState = collections.namedtuple("AttentionState", ("cell_state", "o"))      # attention_cell.py:8
LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


class DeviceState(object):
    """Token for a tensor that lives in the engine's workspace: (region name, time it belongs to, the cell and decode session it came from)."""
    __slots__ = ("region", "time", "owner", "session")

    def __init__(self, region, time, owner=None, session=None):
        self.region, self.time, self.owner, self.session = region, time, owner, session

    def __repr__(self):
        return "<device %s @ step %d>" % (self.region, self.time)


class AttentionCell(object):
    def __init__(self, engine, img, attn_cell_config, num_proj):
        self._engine, self._img = engine, img
        self._dim_e = attn_cell_config["dim_e"]
        self._dim_o = attn_cell_config["dim_o"]
        self._num_units = attn_cell_config["num_units"]
        self._dim_embeddings = attn_cell_config["dim_embeddings"]
        self._num_proj = num_proj
        self._state_size = AttentionState(LSTMStateTuple(self._num_units, self._num_units), self._dim_o)
        self._time = -1            # time of the state the device currently holds (-1: the initial state of the last begin())
        self._session = 0          # bumped by every begin(): tokens of an earlier decode are foreign

    @property
    def state_size(self):
        return self._state_size

    @property
    def output_size(self):
        return self._num_proj

    @property
    def output_dtype(self):
        return "float32"

    @property
    def batch_size(self):
        return int(self._img.shape[0])

    # ---- the device-side session ----
    def begin(self, beam_size=1, max_steps=152, **div):
        """initialize() of the decoder cells: encoder, attention set-up, initial states (attention_cell.py:51-56) for beam_size rows per image."""
        self._engine.decode_begin(self._img, int(beam_size), max_steps=int(max_steps), **div)
        self._session += 1
        self._time = -1
        return self.initial_state()

    def _tokens(self, time):
        mk = lambda region: DeviceState(region, time, self, self._session)
        return AttentionState(LSTMStateTuple(mk("cs"), mk("rec.h")), mk("rec.o"))

    def initial_state(self, time=-1):
        """attention_cell.py:51-56: tanh(mean(img) W_x_0 + b_x_0) for c, h, o -- computed by lxo_decode_begin.  (time >= 0: the tokens
        of the state AFTER step `time`, what the decoder cells hand back from their step().)"""
        return self._tokens(int(time))

    def advance(self, time):
        """The decoder cells' fused step (lxo_decode_step: cell + arg-max / beam bookkeeping) moved the device state to `time`."""
        self._time = int(time)
        return self._tokens(self._time)

    def check_state(self, state):
        """-> "current" for the tokens of the state the device holds, "host" for an AttentionState of arrays; raises on anything else."""
        parts = (state.cell_state.c, state.cell_state.h, state.o)
        if all(isinstance(p, DeviceState) for p in parts):
            for p, region in zip(parts, ("cs", "rec.h", "rec.o")):
                if p.owner is not self or p.session != self._session or p.region != region:
                    raise ValueError("AttentionCell.step: foreign state token %r (another cell, or a decode that has been re-initialised)" % (p,))
                if p.time != self._time:
                    raise ValueError("AttentionCell.step: stale state token %r: the device holds the state of step %d; read a state you want "
                                     "to return to with read_state() and pass the arrays" % (p, self._time))
            return "current"
        if any(isinstance(p, DeviceState) for p in parts):
            raise ValueError("AttentionCell.step: a state mixes device tokens and host arrays")
        return "host"

    def read_state(self, state=None):
        """The state the device holds as host arrays: AttentionState(LSTMStateTuple(c, h), o), float32 [rows, U / U / O]."""
        if state is not None and self.check_state(state) != "current":
            return state
        c, h, o = self._engine.decode_get_state(self._time + 1)
        return AttentionState(LSTMStateTuple(c, h), o)

    def load_state(self, state, embedding=None):
        """Make `state` (host arrays) the state the next step starts from; embedding = the token ids fed with it (None: leave)."""
        c, h, o = (np.ascontiguousarray(a, dtype=np.float32) for a in (state.cell_state.c, state.cell_state.h, state.o))
        ids = None if embedding is None or isinstance(embedding, str) else np.ascontiguousarray(embedding, dtype=np.int32).reshape(-1)
        self._engine.decode_set_state(self._time + 1, c, h, o, ids)

    def step(self, embedding, attn_cell_state):
        """attention_cell.py:58-89: (new_h, new_state-ish) -> here (logits float32 [rows, V], new state tokens).
        embedding: "start_token" (greedy_decoder_cell.py:40-43) or the int32 token ids [rows] whose embeddings are the input -- the lookup
        (greedy_decoder_cell.py:61) happens on the device against the live embedding table; a float array is refused."""
        start = isinstance(embedding, str)
        if start and embedding != "start_token":
            raise ValueError("AttentionCell.step: unknown input %r" % (embedding,))
        if not start:
            e = np.asarray(embedding)
            if e.dtype.kind not in "iu":
                raise TypeError("AttentionCell.step takes token ids (or 'start_token'): the embedding lookup is part of the device-side step")
        kind = self.check_state(attn_cell_state)
        if kind == "host":
            self.load_state(attn_cell_state, None if start else embedding)
        elif not start:
            self._engine.decode_set_state(self._time + 1, None, None, None, np.ascontiguousarray(embedding, dtype=np.int32).reshape(-1))
        logits = self._engine.decode_cell_step(self._time + 1, start)
        self._time += 1
        return logits, self._tokens(self._time)
