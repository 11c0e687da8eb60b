"""Shared by tests/golden/make_ref_decoder_golden.py (build container, writes the fixture) and the tests that read
tests/golden/ref_decoder.npz: the inputs and weights the reference's graph code was run on.

Two weight sets per vocabulary size:
  * "init": the oracle's seed-0 initialisation with the zero-initialised biases replaced by small seeded numbers;
  * "toy":  the same, with a small read-out subset (y_W_o, embeddings, start token, LSTM bias, initial-state biases,
    att_beta; ~35 k numbers, stored in the fixture) trained for a few hundred steps on a toy grammar, so that greedy
    and beam decode produce varied tokens and emit END at staggered steps (random-initialised weights decode to one
    constant token per row and never finish -- a vacuous token-for-token test).  The big tensors stay at their
    seeded values, which is what keeps the fixture small."""
import numpy as np
import torch

from latex_ocr_amd import synthetic
from oracle import ref_model as R

A = "Decoder/AttentionCell/"
TOY_TRAINED = [A + "rnn/y_W_o", "Decoder/embedding_table", "Decoder/start_token", A + "rnn/lstm_cell/bias",
               A + "att_mechanism/b_c_0", A + "att_mechanism/b_h_0", A + "att_mechanism/b_o_0", A + "rnn/att_mechanism/att_beta"]
BIAS_SEED = 11


def perturbed_params(V, seed=0, bias_seed=BIAS_SEED):
    P = R.init_params(V, seed=seed)
    rng = np.random.Generator(np.random.PCG64(bias_seed))
    for k in list(P):
        if k.endswith("/bias"):
            P[k] = torch.from_numpy(rng.uniform(-0.05, 0.05, size=tuple(P[k].shape)).astype(np.float32))
    return P


def toy_params(V, gold):
    """"toy" weights: perturbed_params(V) with the trained read-out subset taken from the fixture."""
    P = perturbed_params(V)
    for k in TOY_TRAINED:
        P[k] = torch.from_numpy(np.array(gold["toyw_v%d__%s" % (V, k.replace("/", "__"))]))
    return P


def toy_set(n, H, W, V, seed):
    """Crops whose ink density (one of four) picks the first token; the rest of the formula walks
    tok -> (5 tok + 3) mod n_ord until a token = 3 (mod 7) (or 14 tokens), so lengths differ by start token."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_ord = V - 3
    inks = [0.03, 0.08, 0.15, 0.25]
    imgs, forms = [], []
    for i in range(n):
        b = int(rng.integers(0, 4))
        im, _ = synthetic.make_set(1, H, W, V, 3, 4, seed=seed * 1000 + i, ink=inks[b])
        tok = (b * 5 + 2) % n_ord
        f = [tok]
        while len(f) < 14 and (tok % 7) != 3:
            tok = (5 * tok + 3) % n_ord
            f.append(tok)
        imgs.append(im[0])
        forms.append(f)
    return imgs, forms


def shape_of(V):
    return (32, 128) if V >= 50 else (32, 48)


def write_mixed_dataset(root, n=23, seed=31):
    """A tiny on-disk dataset in the reference's format with THREE image shapes interleaved (so that same-shape bucketing
    re-orders it) and formulas of 2..9 tokens: PNG images, formulas file, matching file "<img> <formula_idx>"."""
    import os
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = [(8, 16), (8, 24), (12, 16)]
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    with open(os.path.join(root, "formulas.txt"), "w") as ff, open(os.path.join(root, "matching.txt"), "w") as fm:
        for i in range(n):
            h, w = shapes[int(rng.integers(0, 3))]
            img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            Image.fromarray(img).save(os.path.join(root, "images", "%d.png" % i))
            L = int(rng.integers(2, 10))
            ff.write(" ".join("t%d" % int(t) for t in rng.integers(0, 12, size=L)) + "\n")
            fm.write("%d.png %d\n" % (i, i))
    return os.path.join(root, "formulas.txt"), os.path.join(root, "images") + "/", os.path.join(root, "matching.txt")


DATAGEN_CASES = [dict(bucket=False), dict(bucket=True, bucket_size=3), dict(bucket=True, bucket_size=4, max_len=6),
                 dict(bucket=False, max_iter=7, max_len=7), dict(bucket=True, bucket_size=2, iter_mode="full", max_iter=9)]


def run_datagen(DataGenerator, root, greyscale):
    """Drives a DataGenerator class (the reference's or ours) over the mixed dataset; -> JSON-able trace per case."""
    import zlib
    pf, di, pm = write_mixed_dataset(root)
    out = []
    for kw in DATAGEN_CASES:
        ds = DataGenerator(pf, di, pm, img_prepro=greyscale, form_prepro=lambda s: [int(t[1:]) for t in s.strip().split(" ")], **kw)
        items = []
        for inst in ds:
            img, form = inst[0], inst[1]
            rec = {"shape": list(np.asarray(img).shape), "crc": zlib.crc32(np.ascontiguousarray(img).tobytes()) & 0xFFFFFFFF,
                   "dtype": str(np.asarray(img).dtype), "formula": [int(t) for t in form]}
            if len(inst) == 4:
                rec["path"], rec["id"] = inst[2], str(inst[3])
            items.append(rec)
        out.append({"kw": {k: v for k, v in kw.items()}, "items": items, "len": len(ds)})
    return out
