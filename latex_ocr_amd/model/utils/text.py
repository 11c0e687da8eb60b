"""Vocabulary and formula batching (host, integer work).

Mirrors `model/utils/text.py` of the reference (cited per symbol).
"""
from collections import Counter

import numpy as np


def load_tok_to_id(filename, tokens=()):
    """id = line index in the vocab file; `tokens` are appended in order.

    Reference: model/utils/text.py:43-63.  A token repeated in the file keeps
    the LAST line index (dict overwrite), and the specials are numbered from
    len(dict), exactly as the reference does.
    """
    tok_to_id = {}
    with open(filename) as f:
        for idx, line in enumerate(f):
            tok_to_id[line.strip()] = idx
    for tok in tokens:
        tok_to_id[tok] = len(tok_to_id)
    return tok_to_id


def get_form_prepro(vocab, id_unk):
    """formula string -> list of ids, split on single spaces, unknown -> id_unk
    (reference: text.py:26-40)."""
    def prepro(formula):
        return [vocab.get(t, id_unk) for t in formula.strip().split(" ")]
    return prepro


class Vocab(object):
    """Reference: model/utils/text.py:5-23.  Specials are appended in the
    order unk, pad, end => ids V-3, V-2, V-1."""

    def __init__(self, config):
        self.config = config
        self.load_vocab()

    def load_vocab(self):
        special = [self.config.unk, self.config.pad, self.config.end]
        self.tok_to_id = load_tok_to_id(self.config.path_vocab, special)
        self.id_to_tok = {i: t for t, i in self.tok_to_id.items()}
        self.n_tok = len(self.tok_to_id)
        self.id_pad = self.tok_to_id[self.config.pad]
        self.id_end = self.tok_to_id[self.config.end]
        self.id_unk = self.tok_to_id[self.config.unk]

    @property
    def form_prepro(self):
        return get_form_prepro(self.tok_to_id, self.id_unk)


def build_vocab(datasets, min_count=10):
    """Token-count vocabulary over datasets of (img, token list) pairs,
    sorted; tokens seen fewer than `min_count` times are dropped.
    Reference: text.py:93-115."""
    c = Counter()
    for dataset in datasets:
        for _, formula in dataset:
            c.update(formula)
    return sorted(t for t, n in c.items() if n >= min_count)


def write_vocab(vocab, filename):
    """One token per line, no trailing newline (reference: text.py:118-138)."""
    with open(filename, "w") as f:
        f.write("\n".join(vocab))


def pad_batch_formulas(formulas, id_pad, id_end, max_len=None):
    """PAD-fill to the batch max, append END, lengths = len + 1.

    Reference: model/utils/text.py:141-164.
    Returns (int32[B, max_len+1], int32[B]).
    """
    if max_len is None:
        max_len = max(len(f) for f in formulas)
    out = np.full((len(formulas), max_len + 1), id_pad, dtype=np.int32)
    lengths = np.zeros(len(formulas), dtype=np.int32)
    for i, f in enumerate(formulas):
        n = len(f)
        out[i, :n] = np.asarray(f, dtype=np.int32)
        out[i, n] = id_end
        lengths[i] = n + 1
    return out, lengths


def load_formulas(filename):
    """dict line-index -> stripped line (reference: text.py:167-174)."""
    with open(filename) as f:
        return {i: line.strip() for i, line in enumerate(f)}


def beam_backtrace(ids, parents):
    """Follow the beam-search parent pointers: ids, parents int [B, T, k] as lxo_beam_decode returns them ->
    hypotheses int [B, T, k] where out[b, :, i] is the token path that ENDS in beam slot i at the last step.
    The reference's BeamSearchDecoderCell.finalize (beam_search_decoder_cell.py:190-250) never does this (it
    returns ids[:, t, i] per step); this is the optional corrected read-out."""
    import numpy as np
    ids, parents = np.asarray(ids), np.asarray(parents)
    B, T, k = ids.shape
    out = np.empty_like(ids)
    slot = np.tile(np.arange(k)[None, :], (B, 1))
    rows = np.arange(B)[:, None]
    for t in range(T - 1, -1, -1):
        out[:, t, :] = ids[rows, t, slot]
        slot = parents[rows, t, slot]
    return out
