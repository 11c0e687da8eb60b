"""-m gpu: the kept reference surface end to end on the device: synthetic dataset on disk ->
train.py (Config, Vocab, DataGenerator with bucketing, LRSchedule, Img2SeqModel.train with
beam-search validation, checkpoint) -> evaluate_txt.py (auto-restore, write_prediction, metrics)
-> predict_batch; plus the single-rank RCCL path of bench.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_eval_predict(tmp_path, monkeypatch):
    from latex_ocr_amd import synthetic
    monkeypatch.chdir(tmp_path)
    synthetic.write_dataset("data/synthetic", n_train=60, n_val=20, n_test=20)
    os.makedirs("configs")
    for n in ("model.json", "training_small.json", "vocab_small.json", "data_small.json"):
        cfg = json.load(open(os.path.join(ROOT, "configs", n)))
        if n == "training_small.json":
            cfg["n_epochs"] = 2
        json.dump(cfg, open(os.path.join("configs", n), "w"))
    sys.path.insert(0, ROOT)
    import train, evaluate_txt
    best = train.main(["--output", "results/small/"])
    assert np.isfinite(best) and best < 0                       # negated perplexity (quirk C-2)
    wdir = "results/small/model_weights/"
    ck = [f for f in os.listdir(wdir) if f.startswith("model.cpkt-")]
    assert len(ck) == 1 and os.path.exists(wdir + "checkpoint")   # Saver(max_to_keep=1)
    scores = evaluate_txt.main(["--results", "results/small/"])
    assert set(scores) == {"BLEU-4", "ExactMatchScore", "EditDistance", "perplexity"}
    assert os.path.exists("results/small/formulas_test/hyp_1.txt")  # beam_size 2 -> two hypothesis files
    # perplexity after the restore equals what training saw for that checkpoint's weights (same data? no: test set) -> finite
    assert np.isfinite(scores["perplexity"])


def test_single_rank_rccl_path():
    env = dict(os.environ, LXO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                                   "--no-cpu-baseline", "--batch", "8", "--height", "32", "--width", "128", "--vocab", "50"],
                                  env=env, cwd=ROOT, timeout=600)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
