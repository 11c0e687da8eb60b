#!/usr/bin/env python
"""Text evaluation driver with the shape of the reference's evaluate_txt.py:13-50: reload the
configs saved in the results dir, restore the latest checkpoint, decode the test set, score."""
import argparse

from latex_ocr_amd.model.evaluation.text import score_files
from latex_ocr_amd.model.img2seq import Img2SeqModel
from latex_ocr_amd.model.utils.general import Config
from latex_ocr_amd.model.utils.text import Vocab
from train import make_sets


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--results", default="results/small/")
    a = ap.parse_args(argv)
    d = a.results
    config_data, config_vocab, config_model = Config(d + "data.json"), Config(d + "vocab.json"), Config(d + "model.json")
    vocab = Vocab(config_vocab)
    model = Img2SeqModel(config_model, d, vocab)
    model.build_pred()
    (test_set,) = make_sets(config_data, vocab, names=("test",))
    config_eval = Config({"dir_answers": d + "formulas_test/", "batch_size": 20})
    files, perplexity = model.write_prediction(config_eval, test_set)
    scores = score_files(files[0], files[1])
    scores["perplexity"] = perplexity
    model.logger.info("- Test Txt: " + " || ".join("{} is {:04.2f}".format(k, v) for k, v in scores.items()))
    return scores


if __name__ == "__main__":
    main()
