"""Host-side learning-rate schedule.

Mirrors `model/utils/lr_schedule.py:4-118` of the reference: warm start,
per-batch exponential decay between start_decay and end_decay, optional
score-based multiplicative decay and early stopping.  float64 host scalars,
same operation order, so `lr` matches the reference bit for bit.
"""
import numpy as np


class LRSchedule(object):
    def __init__(self, lr_init=1e-3, lr_min=1e-4, start_decay=0, decay_rate=None,
                 end_decay=None, lr_warm=1e-4, end_warm=None, early_stopping=None):
        self._lr_init = lr_init
        self._lr_min = lr_min
        self._start_decay = start_decay
        self._decay_rate = decay_rate
        self._end_decay = end_decay
        self._lr_warm = lr_warm
        self._end_warm = end_warm
        self._score = None
        self._early_stopping = early_stopping
        self._n_batch_no_imprv = 0
        if end_warm is not None:
            # lr_schedule.py:58-60: decay cannot start before the warm-up ends
            self._start_decay = max(end_warm, start_decay)
            self.lr = lr_warm
        else:
            self.lr = lr_init
        if end_decay is not None:
            # lr_schedule.py:64-65
            self._exp_decay = np.power(lr_min / lr_init,
                                       1 / float(end_decay - self._start_decay))

    def state_dict(self):
        """Mutable state (not in the reference, whose Saver never stores it): lr, last score, no-improvement count."""
        return {"lr": float(self.lr), "score": float("nan") if self._score is None else float(self._score),
                "n_batch_no_imprv": float(self._n_batch_no_imprv)}

    def load_state_dict(self, sd):
        self.lr = float(sd["lr"])
        sc = float(sd["score"])
        self._score = None if sc != sc else sc
        self._n_batch_no_imprv = int(float(sd["n_batch_no_imprv"]))

    @property
    def stop_training(self):
        return (self._early_stopping is not None
                and self._n_batch_no_imprv >= self._early_stopping)

    def update(self, batch_no=None, score=None):
        """Reference: lr_schedule.py:82-118 (both updates may fire together)."""
        if batch_no is not None:
            if self._end_warm is not None and self._end_warm <= batch_no <= self._start_decay:
                self.lr = self._lr_init
            if batch_no > self._start_decay and self._end_decay is not None:
                self.lr *= self._exp_decay
        if self._decay_rate is not None and score is not None and self._score is not None:
            if score <= self._score:
                self.lr *= self._decay_rate
                self._n_batch_no_imprv += 1
            else:
                self._n_batch_no_imprv = 0
        if score is not None:
            self._score = score
        self.lr = max(self.lr, self._lr_min)
