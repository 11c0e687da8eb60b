"""TEST INFRASTRUCTURE: tf.contrib of the eager TF stand-in."""
from . import rnn, layers   # noqa: F401
