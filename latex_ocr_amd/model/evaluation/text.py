"""Decode post-processing and text metrics (host).

`truncate_end` / `write_answers` mirror `model/evaluation/text.py:95-145` of
the reference.  The metrics (`score_files`, :12-92) use nltk / `distance` in
the reference; neither is installed, so BLEU-4 (corpus level, nltk's
brevity-penalty definition, no smoothing), Levenshtein and exact match are
restated here in pure Python.
"""
import math
import os
from collections import Counter

from ..utils.general import init_dir
from ..utils.text import load_formulas


def truncate_end(list_of_ids, id_end):
    """Cut at the first END id (reference: evaluation/text.py:95-104)."""
    out = []
    for idx in list_of_ids:
        if idx == id_end:
            break
        out.append(idx)
    return out


def write_answers(references, hypotheses, rev_vocab, dir_name, id_end):
    """ref.txt + one hyp_i.txt per hypothesis rank; ids are cut at END and
    joined with single spaces.  Reference: evaluation/text.py:107-145."""
    def ids_to_str(ids):
        return " ".join(rev_vocab[int(i)] for i in truncate_end(ids, id_end))

    def write_file(name, rows):
        with open(name, "w") as f:
            for r in rows:
                f.write(ids_to_str(r) + "\n")

    init_dir(dir_name)
    names = [dir_name + "ref.txt"]
    write_file(names[0], references)
    for i, hyp in enumerate(hypotheses):
        assert len(references) == len(hyp)
        names.append(dir_name + "hyp_{}.txt".format(i))
        write_file(names[-1], hyp)
    return names


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def exact_match_score(references, hypotheses):
    """Reference: evaluation/text.py:41-56."""
    hits = sum(1 for r, h in zip(references, hypotheses) if list(r) == list(h))
    return hits / float(max(len(hypotheses), 1))


def edit_distance(references, hypotheses):
    """1 - sum(levenshtein) / sum(max len) (reference: evaluation/text.py:75-92)."""
    d, tot = 0, 0.0
    for r, h in zip(references, hypotheses):
        d += levenshtein(r, h)
        tot += float(max(len(r), len(h)))
    return 1.0 - d / tot if tot else 1.0


def _ngrams(seq, n):
    return Counter(tuple(seq[i:i + n]) for i in range(len(seq) - n + 1))


def corpus_bleu(list_of_references, hypotheses, max_n=4):
    """nltk.translate.bleu_score.corpus_bleu(list_of_references, hypotheses, weights = uniform over 1 .. max_n), no smoothing -- the
    published algorithm of the third-party call at evaluation/text.py:70-72 (nltk is not installed here), restated: per n the clipped
    n-gram matches (a hypothesis n-gram counts at most as often as in the reference that has it most) and the hypothesis n-gram totals
    (max(1, .) per sentence, as nltk's `modified_precision` does for hypotheses shorter than n) are summed over the corpus BEFORE the
    division; brevity penalty from the summed hypothesis lengths and the summed closest reference lengths (ties -> the shorter reference).
    Held to nltk's own documented values in tests/test_host_surface.py."""
    num, den = [0] * max_n, [0] * max_n
    hyp_len = ref_len = 0
    for refs, hyp in zip(list_of_references, hypotheses):
        hyp_len += len(hyp)
        ref_len += min((len(r) for r in refs), key=lambda rl: (abs(rl - len(hyp)), rl))
        for n in range(1, max_n + 1):
            h = _ngrams(hyp, n)
            best = Counter()
            for r in refs:
                for g, c in _ngrams(r, n).items():
                    if c > best[g]:
                        best[g] = c
            num[n - 1] += sum(min(c, best[g]) for g, c in h.items())
            den[n - 1] += max(1, len(hyp) - n + 1)
    if num[0] == 0 or min(num) == 0 or min(den) == 0:
        return 0.0
    logp = sum(math.log(n / d) for n, d in zip(num, den)) / max_n
    bp = 1.0 if hyp_len > ref_len else math.exp(1 - ref_len / float(max(hyp_len, 1)))
    return bp * math.exp(logp)


def bleu_score(references, hypotheses, max_n=4):
    """Corpus BLEU-4 with uniform weights, one reference per hypothesis: the call at evaluation/text.py:57-72
    (`references = [[ref] for ref in references]`, then nltk's corpus_bleu)."""
    return corpus_bleu([[ref] for ref in references], hypotheses, max_n)


def score_files(path_ref, path_hyp):
    """Reference: evaluation/text.py:12-38 (tokens split on single spaces)."""
    refs = [r.split(" ") for _, r in load_formulas(path_ref).items()]
    hyps = [h.split(" ") for _, h in load_formulas(path_hyp).items()]
    assert len(refs) == len(hyps)
    return {
        "BLEU-4": bleu_score(refs, hyps) * 100,
        "ExactMatchScore": exact_match_score(refs, hyps) * 100,
        "EditDistance": edit_distance(refs, hyps) * 100,
    }
