"""Synthetic formula-image sets of SURVEY.md section 8(d) (host, NumPy).

No dataset ships with the reference (images and train formulas are missing,
data/.MISSING_LARGE_BLOBS) and there is no network, so every measured or
parity-checked batch is generated here: white (255) background, `ink`
fraction of pixels drawn uniformly from [0, 128); token ids uniform over the
ordinary vocabulary (specials _UNK,_PAD,_END are the last three ids, as
model/utils/text.py:12,60-61 numbers them).
"""
import numpy as np


def make_set(n, H, W, n_tok, len_lo, len_hi, seed=1234, ink=0.08):
    """-> (list of uint8[H,W,1], list of list[int]); lengths ~ U{len_lo..len_hi-1}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    imgs, forms = [], []
    n_ord = n_tok - 3
    for _ in range(n):
        img = np.full((H, W, 1), 255, dtype=np.uint8)
        mask = rng.random((H, W, 1)) < ink
        vals = rng.integers(0, 128, size=(H, W, 1), dtype=np.int64).astype(np.uint8)
        img[mask] = vals[mask]
        L = int(rng.integers(len_lo, len_hi))
        forms.append([int(t) for t in rng.integers(0, n_ord, size=L)])
        imgs.append(img)
    return imgs, forms


# The 21 image sizes of the reference's dataset, as (H, W) after the build-time /2 downsample (configs/data.json:22-28 lists them as
# [W, H] of the rendered PNGs; model/utils/image.py:74-81 `downsample` halves both axes): what DataGenerator's shape buckets
# (data_generator.py:84-122) actually hand to the model, in groups of 20.
REAL_BUCKETS = [(50, 120), (40, 160), (40, 200), (50, 200), (40, 240), (50, 240), (40, 280), (50, 280), (40, 320), (50, 320), (40, 360),
                (50, 360), (60, 360), (100, 360), (50, 400), (160, 400), (100, 500), (200, 500), (100, 600), (100, 800), (800, 800)]


def bucket_lengths(W):
    """Formula-length range used with a bucket of width W in the synthetic sweeps: a rendered formula is about as long as it is wide
    (~4 .. 16 pixels per token after the downsample), capped by max_length_formula = 150 (configs/data.json:20)."""
    return max(5, W // 16), min(150, max(8, W // 4)) + 1


def config1(seed=1234):
    """100 crops 32x128, vocab 50, lengths U{5..20} (BASELINE.json configs[0])."""
    return make_set(100, 32, 128, 50, 5, 21, seed)


def config3_batch(B=64, H=128, W=512, n_tok=500, seed=1234):
    """One training batch of BASELINE.json configs[2]: lengths U{30..100}."""
    return make_set(B, H, W, n_tok, 30, 101, seed)


def write_dataset(root, n_train=100, n_val=20, n_test=20, H=32, W=128, n_ord=47, seed=1234):
    """Materialise a synthetic dataset in the reference's on-disk format (PNG images, a formulas
    file of space-separated tokens, a matching file of "<img> <formula_idx>" lines, vocab.txt) so
    that DataGenerator / Vocab / train.py can be driven end to end.  Tokens are "t0".."t<n_ord-1>"."""
    import os
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    toks = ["t%d" % i for i in range(n_ord)]
    with open(os.path.join(root, "vocab.txt"), "w") as f:
        f.write("\n".join(toks))
    for split, n, s in (("train", n_train, seed), ("val", n_val, seed + 1), ("test", n_test, seed + 2)):
        imgs, forms = make_set(n, H, W, n_ord + 3, 5, 21, seed=s)
        d = os.path.join(root, split)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(root, split + ".formulas.txt"), "w") as ff, open(os.path.join(root, split + ".matching.txt"), "w") as fm:
            for i, (img, form) in enumerate(zip(imgs, forms)):
                Image.fromarray(np.repeat(img, 3, axis=2)).save(os.path.join(d, "%d.png" % i))
                ff.write(" ".join(toks[t] for t in form) + "\n")
                fm.write("%d.png %d\n" % (i, i))
    return root
