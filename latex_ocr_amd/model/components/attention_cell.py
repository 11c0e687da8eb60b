"""AttentionCell of the reference (model/components/attention_cell.py:11-102) as a handle on the device-side cell.

In the reference the cell owns the graph of one step: LSTMCell, dropout, AttentionMechanism.context, the o / logits
projections (attention_cell.py:58-89).  Here that step is one call of the HIP library (lxo_decode_step runs
csrc/model_decoder.hip decode_common_step: the fused LSTM / attention / tanh-o kernels + the logits GEMM) and the state
(AttentionState(cell_state=LSTMStateTuple(c, h), o)) never leaves the GPU, so the object carries the image batch and the
hyper-parameters and hands out opaque state tokens."""
import collections

AttentionState = collections.namedtuple("AttentionState", ("cell_state", "o"))      # attention_cell.py:8
LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


class DeviceState(object):
    """Opaque token for a tensor that lives in the engine's workspace (region name, time it belongs to)."""
    __slots__ = ("region", "time")

    def __init__(self, region, time):
        self.region, self.time = region, time

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

    def initial_state(self, time=-1):
        """attention_cell.py:51-56: tanh(mean(img) W_x_0 + b_x_0) for c, h, o -- computed by lxo_decode_begin."""
        return AttentionState(LSTMStateTuple(DeviceState("cs", time), DeviceState("rec.h", time)), DeviceState("rec.o", time))
