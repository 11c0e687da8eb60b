#!/usr/bin/env python
"""Training driver with the shape of the reference's train.py:23-61 (json configs -> Vocab,
DataGenerator x2, LRSchedule in batches, Img2SeqModel.build_train / train)."""
import argparse

from latex_ocr_amd.model.img2seq import Img2SeqModel
from latex_ocr_amd.model.utils.data_generator import DataGenerator
from latex_ocr_amd.model.utils.general import Config
from latex_ocr_amd.model.utils.image import greyscale
from latex_ocr_amd.model.utils.lr_schedule import LRSchedule
from latex_ocr_amd.model.utils.text import Vocab


def make_sets(config, vocab, names=("train", "val")):
    out = []
    for n in names:
        out.append(DataGenerator(path_formulas=getattr(config, "path_formulas_" + n), dir_images=getattr(config, "dir_images_" + n),
                                 img_prepro=greyscale, max_iter=config.max_iter, bucket=getattr(config, "bucket_" + n),
                                 path_matching=getattr(config, "path_matching_" + n), max_len=config.max_length_formula,
                                 form_prepro=vocab.form_prepro))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="configs/data_small.json")
    ap.add_argument("--vocab", default="configs/vocab_small.json")
    ap.add_argument("--training", default="configs/training_small.json")
    ap.add_argument("--model", default="configs/model.json")
    ap.add_argument("--output", default="results/small/")
    a = ap.parse_args(argv)
    config = Config([a.data, a.vocab, a.training, a.model])
    config.save(a.output)
    vocab = Vocab(config)
    train_set, val_set = make_sets(config, vocab)
    n_batches_epoch = (len(train_set) + config.batch_size - 1) // config.batch_size
    lr_schedule = LRSchedule(lr_init=config.lr_init, start_decay=config.start_decay * n_batches_epoch,
                             end_decay=config.end_decay * n_batches_epoch, end_warm=config.end_warm * n_batches_epoch,
                             lr_warm=config.lr_warm, lr_min=config.lr_min)
    model = Img2SeqModel(config, a.output, vocab)
    model.build_train(config)
    return model.train(config, train_set, val_set, lr_schedule)


if __name__ == "__main__":
    main()
