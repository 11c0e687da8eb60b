"""TEST INFRASTRUCTURE: tensorflow.python.ops.rnn (imported by dynamic_decode.py:3, not used)."""
from ...nn import dynamic_rnn   # noqa: F401
