"""The reference's decoder-cell protocol (model/components/{dynamic_decode,greedy_decoder_cell,beam_search_decoder_cell,
attention_cell}.py) over the step-wise C-ABI calls lxo_decode_begin / lxo_decode_step: same class and method names, same
return structures; the tensors behind them live in the engine's workspace on the GPU."""
from .dynamic_decode import dynamic_decode, transpose_batch_time                       # noqa: F401
from .greedy_decoder_cell import GreedyDecoderCell, DecoderOutput                      # noqa: F401
from .beam_search_decoder_cell import BeamSearchDecoderCell, BeamSearchDecoderOutput, BeamSearchDecoderCellState   # noqa: F401
from .attention_cell import AttentionCell, AttentionState                             # noqa: F401
