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
    # the same weights through a tf.train.Saver bundle (the reference's container): predictions and attention identical
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_checkpoint as CC
    from latex_ocr_amd.model.img2seq import Img2SeqModel
    from latex_ocr_amd.model.utils.general import Config
    from latex_ocr_amd.model.utils.image import greyscale
    from latex_ocr_amd.model.utils.text import Vocab
    from PIL import Image
    os.makedirs("results/tf/model_weights")
    for n in ("model.json", "vocab.json"):
        shutil.copy("results/small/" + n, "results/tf/" + n)
    CC.main([wdir + ck[0], "results/tf/model_weights/" + ck[0], "--to", "tf"])
    open("results/tf/model_weights/checkpoint", "w").write('model_checkpoint_path: "%s"\n' % ck[0])
    img_path = sorted(p for p in os.listdir("data/synthetic/test") if p.endswith(".png"))[0]
    img = greyscale(np.asarray(Image.open("data/synthetic/test/" + img_path).convert("RGB")))
    hyps = []
    for d in ("results/small/", "results/tf/"):
        m = Img2SeqModel(Config(d + "model.json"), d, Vocab(Config(d + "vocab.json")))
        m.build_pred()
        hyps.append((m.predict(img), m.predict_with_attention(img)))
    assert hyps[0][0] == hyps[1][0] and hyps[0][1][0] == hyps[1][1][0]
    # the attention path honours config.decoding (the shipped model.json decodes with beam_search, k = 2): the same best hypothesis as
    # predict(), and one attention slice per STEP of the loop -- it runs until every beam has finished, so >= tokens + END
    assert m._config.decoding == "beam_search" and hyps[1][1][0] == hyps[1][0][0]
    assert hyps[1][1][1].ndim == 3 and hyps[1][1][1].shape[0] >= len(hyps[1][1][0].split()) and abs(float(hyps[1][1][1][0].sum()) - 1.0) < 1e-2
    assert np.array_equal(hyps[0][1][1], hyps[1][1][1])
    import visualize_attention as VA
    hyp, files = VA.vis_img_with_attention(m, "data/synthetic/test/" + img_path, "results/tf/")
    assert all(os.path.exists(f) for f in files)


@pytest.mark.parametrize("host_ordered,comm", [("1", "abi"), ("0", "abi"), ("1", "torch")])
def test_single_rank_rccl_path(host_ordered, comm):
    """bench.py's data-parallel path at world size 1 on RCCL: through the C ABI (lxo_comm_init / lxo_allreduce_bucket; the default) with
    buckets ordered by the helper thread and by stream waits, and through torch.distributed's nccl group (LXO_DP_COMM=torch)"""
    env = dict(os.environ, LXO_FORCE_DIST="1", LXO_DP_HOST_ORDERED=host_ordered, LXO_DP_COMM=comm, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                                   "--no-cpu-baseline", "--no-pmc", "--batch", "8", "--height", "32", "--width", "128", "--vocab", "50"],
                                  env=env, cwd=ROOT, timeout=600)
    line = [l for l in out.decode().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"
    assert d["data_parallel"]["exposed_allreduce_ms_per_step"][0] is not None
    assert d["data_parallel"]["rccl_ranks_seen"] == (1 if comm == "abi" else None), d["data_parallel"]
