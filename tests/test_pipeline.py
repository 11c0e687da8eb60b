"""CPU: input pipeline (latex_ocr_amd/pipeline.py) -- prefetch order/contents and shape-bucketed DP sharding."""
import numpy as np

from latex_ocr_amd import synthetic
from latex_ocr_amd.model.utils.data_generator import ListDataset
from latex_ocr_amd.model.utils.general import minibatches
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
from latex_ocr_amd.pipeline import Prefetcher, ShardedBuckets


def _mixed_set():
    a_i, a_f = synthetic.make_set(23, 32, 64, 30, 3, 9, seed=1)
    b_i, b_f = synthetic.make_set(17, 40, 96, 30, 3, 9, seed=2)
    imgs, forms = [], []
    for i in range(23):                      # interleave two shapes
        imgs.append(a_i[i]); forms.append(a_f[i])
        if i < 17:
            imgs.append(b_i[i]); forms.append(b_f[i])
    return ListDataset(imgs, forms)


def test_prefetcher_matches_synchronous_feed():
    ds = _mixed_set()
    got = list(Prefetcher(ds, 8, id_pad=28, id_end=29, device="cpu", depth=3))
    ref = list(minibatches(ds, 8))
    assert len(got) == len(ref) == 5
    for b, (imgs, forms) in zip(got, ref):
        f, l = pad_batch_formulas(forms, 28, 29)
        assert np.array_equal(b.img.numpy(), pad_batch_images(imgs))
        assert np.array_equal(b.formula.numpy(), f) and np.array_equal(b.lengths, l)
        assert b.n_tokens == int(l.sum()) and b.size == len(imgs)


def test_prefetcher_propagates_loader_errors_and_can_stop_early():
    class Bad(object):
        def __iter__(self):
            yield np.zeros((8, 8, 1), np.uint8), [1, 2]
            yield np.zeros((8, 8, 1), np.uint8), [1, 2]
            raise RuntimeError("decode failed")
    it = iter(Prefetcher(Bad(), 1, 0, 1, device="cpu"))
    next(it)
    try:
        next(it)
        assert False
    except RuntimeError as e:
        assert "decode failed" in str(e)
    it2 = iter(Prefetcher(_mixed_set(), 2, 28, 29, device="cpu", depth=1))
    next(it2)
    it2.close()                               # generator finaliser must not hang on the full queue


def test_sharded_buckets_equal_shapes_and_full_coverage():
    ds = _mixed_set()
    world, bs = 2, 4
    per_rank = [list(ShardedBuckets(ds, bs, world, r)) for r in range(world)]
    assert len(per_rank[0]) == len(per_rank[1])
    seen = 0
    for (i0, f0), (i1, f1) in zip(*per_rank):
        assert len({im.shape for im in i0 + i1}) == 1          # one image shape per step on every rank
        assert len(i0) >= 1 and len(i1) >= 1 and abs(len(i0) - len(i1)) <= 1
        seen += len(i0) + len(i1)
    sb = ShardedBuckets(ds, bs, world, 0); list(sb)
    assert seen + sb.dropped == len(ds)
    # the step count the LR schedule is scaled with = what a pass really yields (two shape buckets: 3 steps for 23 crops + 2 for 17, whose 1-crop tail is dropped)
    assert len(sb) == len(per_rank[0]) == 5 and len(ShardedBuckets(ds, bs, 1, 0)) == len(list(ShardedBuckets(ds, bs, 1, 0)))
    # world 1 = the reference's bucket grouping, nothing dropped
    one = list(ShardedBuckets(ds, bs, 1, 0))
    assert sum(len(b[0]) for b in one) == len(ds)
