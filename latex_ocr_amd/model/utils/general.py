"""Host-side helpers kept from the reference surface.

Mirrors `model/utils/general.py` of the reference (cited per symbol); written
from its behaviour, not its text.  Nothing here touches the device.
"""
import json
import logging
import os
import shutil
import sys
import time


def minibatches(data_generator, minibatch_size):
    """Group consecutive (img, formula) pairs into lists of `minibatch_size`.

    Reference: model/utils/general.py:15-35.  The last batch may be short;
    an empty generator yields nothing.
    """
    imgs, forms = [], []
    for img, form in data_generator:
        if len(imgs) == minibatch_size:
            yield imgs, forms
            imgs, forms = [], []
        imgs.append(img)
        forms.append(form)
    if imgs:
        yield imgs, forms


def init_dir(dir_name):
    """mkdir -p (reference: general.py:63-67)."""
    if dir_name is not None and not os.path.exists(dir_name):
        os.makedirs(dir_name)


def get_logger(filename):
    """stdout + file logger (reference: general.py:50-60)."""
    logger = logging.getLogger("logger")
    logger.setLevel(logging.INFO)
    logging.basicConfig(format="%(message)s", level=logging.INFO)
    handler = logging.FileHandler(filename)
    handler.setLevel(logging.INFO)
    handler.setFormatter(logging.Formatter("%(asctime)s:%(levelname)s: %(message)s"))
    logging.getLogger().addHandler(handler)
    return logger


class Config(object):
    """Flat attribute namespace loaded from a dict, a json path, or a list of
    json paths merged in order (later files override earlier keys).

    Reference: model/utils/general.py:88-120.  `save` copies each source json
    under its own `export_name`; dict sources write nothing (quirk C-9).
    """

    def __init__(self, source):
        self.source = source
        if isinstance(source, dict):
            self.__dict__.update(source)
        elif isinstance(source, (list, tuple)):
            for s in source:
                self.load_json(s)
        else:
            self.load_json(source)

    def load_json(self, source):
        with open(source) as f:
            self.__dict__.update(json.load(f))

    def save(self, dir_name):
        init_dir(dir_name)
        if isinstance(self.source, (list, tuple)):
            for s in self.source:
                Config(s).save(dir_name)
        elif isinstance(self.source, dict):
            pass
        else:
            shutil.copyfile(self.source, os.path.join(dir_name, self.export_name))


class Progbar(object):
    """Running-mean progress line (reference: general.py:123-223).  Console
    cosmetics are reduced to one carriage-returned line; the averaged values
    and the `info` string (logged at epoch end, img2seq.py:184) are kept."""

    def __init__(self, max_step, width=30, stream=None):
        self.max_step = max_step
        self.width = width
        self.sum_values = {}
        self.start = time.time()
        self.last_step = 0
        self.info = ""
        self._stream = stream if stream is not None else sys.stdout

    def update(self, curr_step, values):
        dn = curr_step - self.last_step
        for k, v in values:
            acc = self.sum_values.setdefault(k, [0.0, 0])
            acc[0] += v * dn
            acc[1] += dn
        now = time.time()
        if curr_step < self.max_step:
            per = (now - self.start) / curr_step if curr_step else 0.0
            info = " - ETA: %ds" % (per * (self.max_step - curr_step))
        else:
            info = " - %ds" % (now - self.start)
        for name, (tot, n) in self.sum_values.items():
            info += " - %s: %.6f" % (name, tot / max(1, n))
        self.info = info
        done = int(self.width * float(curr_step) / max(1, self.max_step))
        bar = "%d/%d [%s%s]" % (curr_step, self.max_step, "=" * done, "." * (self.width - done))
        self._stream.write("\r" + bar + info)
        if curr_step >= self.max_step:
            self._stream.write("\n")
        self._stream.flush()
        self.last_step = curr_step
