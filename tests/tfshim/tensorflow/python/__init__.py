"""TEST INFRASTRUCTURE: tensorflow.python of the eager TF stand-in."""
from . import util, ops   # noqa: F401
