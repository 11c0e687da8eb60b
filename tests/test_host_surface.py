"""CPU: the kept host surface (Config / Vocab / batching / LRSchedule) against golden vectors
produced by the reference's own modules (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

from latex_ocr_amd.model.utils import general as G, text as T, image as I
from latex_ocr_amd.model.utils.lr_schedule import LRSchedule

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_surface.json")))


def test_minibatches():
    data = [(i, [i, i + 1]) for i in range(11)]
    for bs, want in GOLD["minibatches"].items():
        got = [[list(x), list(y)] for x, y in G.minibatches(iter(data), int(bs))]
        assert got == want
    assert list(G.minibatches(iter([]), 4)) == []


def test_pad_batch_formulas():
    g = GOLD["pad_batch_formulas"]
    f, l = T.pad_batch_formulas(g["in"], 48, 49)
    assert f.tolist() == g["out"] and l.tolist() == g["len"] and str(f.dtype) == g["dtype"]


def test_pad_batch_images_and_greyscale():
    g = GOLD["pad_batch_images"]
    out = I.pad_batch_images([np.asarray(a, np.uint8) for a in g["in"]])
    assert out.tolist() == g["out"] and str(out.dtype) == g["dtype"]
    g = GOLD["greyscale"]
    assert I.greyscale(np.asarray(g["in"], np.uint8)).tolist() == g["out"]


def test_vocab(tmp_path):
    g = GOLD["vocab"]
    vp = str(tmp_path / "vocab.txt")
    T.write_vocab(g["tokens"], vp)
    assert open(vp).read() == g["file"]
    v = T.Vocab(G.Config({"unk": "_UNK", "pad": "_PAD", "end": "_END", "path_vocab": vp}))
    assert v.tok_to_id == g["tok_to_id"] and v.n_tok == g["n_tok"]
    assert (v.id_unk, v.id_pad, v.id_end) == (g["id_unk"], g["id_pad"], g["id_end"]) == (v.n_tok - 3, v.n_tok - 2, v.n_tok - 1)
    assert v.form_prepro(g["prepro_in"]) == g["prepro_out"]


def test_build_vocab():
    ds = [(None, ["a", "b", "a"]), (None, ["b", "c", "a"])]
    assert T.build_vocab([ds], min_count=2) == ["a", "b"]


def test_lr_schedule_traces():
    for tr in GOLD["lr_schedule"]:
        s = LRSchedule(**tr["kw"])
        lrs = [s.lr]
        scores = [-3.0, -2.5, -2.6, -2.7, -2.4, -2.9] if tr["kw"].get("decay_rate") else None
        for i in range(len(tr["lrs"]) - 1):
            s.update(batch_no=i)
            if scores is not None and i % 5 == 4:
                s.update(score=scores[i // 5])
            lrs.append(s.lr)
        assert lrs == tr["lrs"]            # bit-for-bit (same float64 operation order)
        assert bool(s.stop_training) == tr["stop"]


def test_config_merge(tmp_path):
    a, b = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    json.dump({"export_name": "a.json", "x": 1, "y": 2}, open(a, "w"))
    json.dump({"export_name": "b.json", "y": 3, "z": {"k": 4}}, open(b, "w"))
    c = G.Config([a, b])
    assert {"x": c.x, "y": c.y, "z": c.z, "export_name": c.export_name} == GOLD["config_merge"]
    c.save(str(tmp_path / "out") + "/")
    assert sorted(os.listdir(str(tmp_path / "out"))) == ["a.json", "b.json"]


def test_eval_text():
    from latex_ocr_amd.model.evaluation import text as E
    assert E.truncate_end([3, 4, 9, 5, 9], 9) == [3, 4]
    assert E.levenshtein("kitten", "sitting") == 3
    refs, hyps = [["a", "b", "c", "d", "e"], ["x", "y"]], [["a", "b", "c", "d", "e"], ["x", "z"]]
    assert E.exact_match_score(refs, hyps) == 0.5
    assert abs(E.edit_distance(refs, hyps) - (1 - 1 / 7.0)) < 1e-12
    assert E.bleu_score([["a", "b", "c", "d", "e"]], [["a", "b", "c", "d", "e"]]) == 1.0
    # nltk corpus_bleu semantics for hypotheses shorter than n: modified_precision returns Fraction(num, max(1, sum(counts))),
    # i.e. every short hypothesis still adds 1 to the n-gram denominator.  Hand computation for the two pairs below:
    #   p1 = (5+2)/(5+2), p2 = (4+1)/(4+1), p3 = 3/(3+1), p4 = 2/(2+1); lengths equal -> BP = 1
    refs = [["a", "b", "c", "d", "e"], ["x", "y"]]
    hyps = [["a", "b", "c", "d", "e"], ["x", "y"]]
    want = (1.0 * 1.0 * (3 / 4.0) * (2 / 3.0)) ** 0.25
    assert abs(E.bleu_score(refs, hyps) - want) < 1e-12


def test_beam_backtrace_follows_parents():
    from latex_ocr_amd.model.utils.text import beam_backtrace
    # two beams, three steps: at t=1 both slots extend slot 0; at t=2 slot 0 extends slot 1 and slot 1 extends slot 0
    ids = np.array([[[5, 6], [7, 8], [9, 3]]])
    par = np.array([[[0, 0], [0, 0], [1, 0]]])
    out = beam_backtrace(ids, par)
    assert out[0, :, 0].tolist() == [5, 8, 9]
    assert out[0, :, 1].tolist() == [5, 7, 3]


def test_encoder_out_hw_and_attention_overlay(tmp_path):
    from PIL import Image
    from latex_ocr_amd.model.utils.image import encoder_out_hw
    import visualize_attention as VA
    assert encoder_out_hw(128, 512) == (14, 62) and encoder_out_hw(40, 240) == (3, 28)
    assert VA.getWH(512, 128) == (62, 14)
    img = Image.fromarray((np.random.RandomState(0).rand(40, 240) * 255).astype(np.uint8))
    aw, ah = VA.getWH(240, 40)
    alphas = np.random.RandomState(1).dirichlet(np.ones(aw * ah), size=3).reshape(3, ah, aw)
    arr = VA.getOutArray(alphas[0], aw, ah)
    assert arr.shape == (ah, aw) and abs(arr[1, 2] - (1 - alphas[0, 1, 2]) * 255) < 1e-9
    files = VA.vis_attention_slices(img, alphas, str(tmp_path / "vis"))
    gif = VA.vis_attention_gif(img, alphas, str(tmp_path / "vis"), "a b c")
    assert len(files) == 3 and all(os.path.exists(f) for f in files + [gif])
    assert Image.open(files[0]).size == (240, 40)


def test_data_generator_vs_reference_class(tmp_path):
    """DataGenerator (matching file, greyscale prepro, max_len filter, max_iter, same-shape bucketing in groups of bucket_size,
    "full" iteration mode, len()) against a trace of the REFERENCE's own class (data_generator.py:35-215) over the same seeded
    on-disk dataset (tests/golden/make_ref_datagen_golden.py; scipy.misc.imread supplied by PIL)."""
    import json
    import refgold
    from latex_ocr_amd.model.utils.data_generator import DataGenerator
    from latex_ocr_amd.model.utils.image import greyscale
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "data_generator.json")))
    got = refgold.run_datagen(DataGenerator, str(tmp_path), greyscale)
    assert len(got) == len(want) == len(refgold.DATAGEN_CASES)
    for g, w in zip(got, want):
        assert g["kw"] == w["kw"] and g["len"] == w["len"], (g["kw"], g["len"], w["len"])
        assert g["items"] == w["items"], g["kw"]


def test_text_metrics_vs_the_published_values_of_the_third_party_libraries():
    """model/evaluation/text.py:57-92 delegates BLEU to nltk (`nltk.translate.bleu_score.corpus_bleu`) and the edit distance to the `distance`
    package; neither is installed here, so both are restated (latex_ocr_amd/model/evaluation/text.py).  What pins the restatements: the
    known answers those libraries PUBLISH -- the doctest values in nltk's bleu_score.py (sentence_bleu 0.5045..., the mean of two sentence
    scores 0.6223..., corpus_bleu of the two-sentence corpus 0.5920...: same sentences, one to three references, default uniform 4-gram
    weights, no smoothing) and the examples of the `distance` README (levenshtein("lenvestein", "levenshtein") == 3)."""
    from latex_ocr_amd.model.evaluation.text import corpus_bleu, bleu_score, levenshtein, edit_distance, exact_match_score
    hyp1 = "It is a guide to action which ensures that the military always obeys the commands of the party".split()
    ref1a = "It is a guide to action that ensures that the military will forever heed Party commands".split()
    ref1b = "It is the guiding principle which guarantees the military forces always being under the command of the Party".split()
    ref1c = "It is the practical guide for the army always to heed the directions of the party".split()
    hyp2 = "he read the book because he was interested in world history".split()
    ref2a = "he was interested in world history because he read the book".split()
    s1 = corpus_bleu([[ref1a, ref1b, ref1c]], [hyp1])             # a one-sentence corpus = nltk's sentence_bleu
    s2 = corpus_bleu([[ref2a]], [hyp2])
    # (the doctests print the value and elide the rest: "0.5045..." = a prefix of repr(value))
    assert repr(s1).startswith("0.5045")
    assert repr((s1 + s2) / 2).startswith("0.6223")
    assert repr(corpus_bleu([[ref1a, ref1b, ref1c], [ref2a]], [hyp1, hyp2])).startswith("0.5920")
    assert bleu_score([ref2a], [hyp2]) == s2                      # the reference's call shape: one reference per hypothesis (evaluation/text.py:69-71)
    assert levenshtein("lenvestein", "levenshtein") == 3 and levenshtein("kitten", "sitting") == 3 and levenshtein([], [1, 2]) == 2
    assert abs(edit_distance([list("kitten")], [list("sitting")]) - (1 - 3 / 7.0)) < 1e-12
    assert exact_match_score([[1, 2], [3]], [[1, 2], [4]]) == 0.5
