#!/usr/bin/env python
"""tests/golden/data_generator.json from the REFERENCE's own DataGenerator (model/utils/data_generator.py:35-215): the
module imports once `scipy.misc.imread` (removed from SciPy >= 1.2) is supplied -- here by PIL, which is what SciPy 1.0's
imread called underneath.  Build container only; the test reads the committed JSON and re-creates the same seeded dataset.

    python tests/golden/make_ref_datagen_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import scipy                                                     # noqa: E402
misc = types.ModuleType("scipy.misc")


def imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


misc.imread = imread
sys.modules["scipy.misc"] = misc
scipy.misc = misc
from model.utils.data_generator import DataGenerator             # noqa: E402   (the reference's class)
from model.utils.image import greyscale                          # noqa: E402
import refgold                                                   # noqa: E402

with tempfile.TemporaryDirectory() as d:
    trace = refgold.run_datagen(DataGenerator, d, greyscale)
json.dump(trace, open(os.path.join(HERE, "data_generator.json"), "w"), indent=0)
print("wrote data_generator.json:", [(c["kw"], len(c["items"]), c["len"]) for c in trace])
