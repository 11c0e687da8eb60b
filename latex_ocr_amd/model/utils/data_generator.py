"""Dataset iterator kept from the reference surface (host only).

Mirrors `model/utils/data_generator.py:35-238`: an iterable over
(image, token ids) read from a matching file of "<img> <formula_idx>" lines,
with same-shape bucketing in groups of `bucket_size`, a max-length filter and
`max_iter`.  `scipy.misc.imread` (removed from SciPy) is replaced by PIL.
`build()` (LaTeX rendering) is out of scope.
"""
import numpy as np

from .text import load_formulas


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path))


class DataGeneratorFile(object):
    """(img_path, formula_id) pairs from a matching file
    (reference: data_generator.py:11-32)."""

    def __init__(self, filename):
        self._filename = filename

    def __iter__(self):
        with open(self._filename) as f:
            for line in f:
                parts = line.strip().split(" ")
                yield parts[0], parts[1]


class DataGenerator(object):
    def __init__(self, path_formulas, dir_images, path_matching, bucket=False,
                 form_prepro=lambda s: s.strip().split(" "), iter_mode="data",
                 img_prepro=lambda x: x, max_iter=None, max_len=None,
                 bucket_size=20, imread=_imread, reference_bucket_quirk=True):
        self._path_formulas = path_formulas
        self._dir_images = dir_images
        self._path_matching = path_matching
        self._img_prepro = img_prepro
        self._form_prepro = form_prepro
        self._max_iter = max_iter
        self._max_len = max_len
        self._iter_mode = iter_mode
        self._bucket = bucket
        self._bucket_size = bucket_size
        self._imread = imread
        self._reference_bucket_quirk = bool(reference_bucket_quirk)
        self._length = None
        self._formulas = load_formulas(path_formulas)
        self._data_generator = DataGeneratorFile(path_matching)
        if bucket:
            self._data_generator = self.bucket(bucket_size)

    def bucket(self, bucket_size):
        """One full pass; emit groups of `bucket_size` same-shape images as they fill, leftovers at the end in first-seen
        shape order.  Reference: data_generator.py:84-122 (also sets the dataset length).
        Reference quirk, found by running the reference's own class (tests/golden/make_ref_datagen_golden.py) and kept by
        default: its flush loop `for (img_path, formula_id) in data_buckets[s]` (:107-108) re-binds the names of the CURRENT
        example, so the example that arrives at a full bucket is lost and the LAST element of the flushed bucket opens the new
        one instead (one duplicate + one dropped sample per flush).  reference_bucket_quirk=False buckets without it."""
        out, buckets = [], {}
        old_mode, self._iter_mode = self._iter_mode, "full"
        n = 0
        for img, _, img_path, formula_id in self:
            n += 1
            group = buckets.setdefault(img.shape, [])
            cur = (img_path, formula_id)
            if len(group) == bucket_size:
                out.extend(group)
                if self._reference_bucket_quirk:
                    cur = group[-1]
                group.clear()
            group.append(cur)
        for group in buckets.values():
            out.extend(group)
        self._iter_mode = old_mode
        self._length = n
        return out

    def _process_instance(self, example):
        """Reference: data_generator.py:147-179."""
        img_path, formula_id = example
        img = self._img_prepro(self._imread(self._dir_images + img_path))
        formula = self._form_prepro(self._formulas[int(formula_id)])
        inst = (img, formula) if self._iter_mode == "data" else (img, formula, img_path, formula_id)
        skip = self._max_len is not None and len(formula) > self._max_len
        return inst, skip

    def __iter__(self):
        n_iter = 0
        for example in self._data_generator:
            if self._max_iter is not None and n_iter >= self._max_iter:
                break
            result, skip = self._process_instance(example)
            if skip:
                continue
            n_iter += 1
            yield result

    def __len__(self):
        if self._length is None:
            self._length = sum(1 for _ in self)
        return self._length


class ListDataset(object):
    """In-memory dataset of (uint8[H,W,1], list[int]) pairs with the same
    iteration protocol; used for the synthetic sets of SURVEY.md section 8(d)."""

    def __init__(self, images, formulas):
        assert len(images) == len(formulas)
        self.images, self.formulas = images, formulas

    def __iter__(self):
        return iter(zip(self.images, self.formulas))

    def __len__(self):
        return len(self.images)
