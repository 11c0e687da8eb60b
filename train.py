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
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:        # one process per GPU: python -m torch.distributed.run --nproc-per-node N train.py ...
        import torch
        import torch.distributed as td
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(local)
            config.device = "cuda:%d" % local
        # control plane (the RCCL id, barriers, end-of-epoch scalars) on gloo; the gradient buckets travel on RCCL through liblxo's C ABI
        td.init_process_group(backend="nccl" if (use_gpu and os.environ.get("LXO_DP_COMM", "abi") == "torch") else "gloo")
        from latex_ocr_amd.dist import DataParallel
        dist = DataParallel(device=config.device if use_gpu else "cpu")
        # steps per epoch under data parallelism = what ShardedBuckets will yield (per shape bucket), not ceil(N / (bs * world))
        from latex_ocr_amd.pipeline import ShardedBuckets
        n_batches_epoch = len(ShardedBuckets(train_set, config.batch_size, world, int(os.environ.get("RANK", "0"))))
        lr_schedule = LRSchedule(lr_init=config.lr_init, start_decay=config.start_decay * n_batches_epoch,
                                 end_decay=config.end_decay * n_batches_epoch, end_warm=config.end_warm * n_batches_epoch,
                                 lr_warm=config.lr_warm, lr_min=config.lr_min)
    model = Img2SeqModel(config, a.output, vocab)
    model.build_train(config)
    if dist is not None:
        model.attach_dist(dist)
    best = model.train(config, train_set, val_set, lr_schedule)
    if dist is not None:
        dist.barrier()
        dist.close()                     # the helper thread of the host-ordered buckets, the RCCL communicator
        td.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
