from . import rnn   # noqa: F401
