"""The training step over the reference's 21 real image sizes (configs/data.json:22-28, halved) at batch 3, 20 and 64 -> a table on stdout and
gpurun_out/<tag>_buckets.json (bench.py: real_buckets).  python tools/real_buckets.py [tag] [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batches = tuple(int(x) for x in os.environ.get("RB_BATCHES", "3,20,64").split(","))
print("%4s %4s %3s %3s %4s %6s | %8s %9s %8s | %6s %6s | %7s %7s" % ("H", "W", "B", "Bc", "T", "R", "ms/step", "img/s", "us/img", "conv", "wgrad", "fwd us", "bwd us"))


def log(r):
    if "error" in r:
        print(r, flush=True)
        return
    print("%4d %4d %3d %3d %4d %6d | %8.3f %9.1f %8.1f | %6s %6s | %7s %7s  %s" % (
        r["H"], r["W"], r["B"], r["chain_batch"], r["T"], r["regions"], r["ms_per_step"], r["img_per_s"], r["us_per_image"],
        r["conv_fwd_dgrad_frac"], r["conv_wgrad_frac"], r["chain_fwd_us_per_step"], r["chain_bwd_us_per_step"], "" if r["chains"] else "NO CHAIN"), flush=True)


shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ["RB_SHAPES"].split(",")] if os.environ.get("RB_SHAPES") else None
rows = bench.real_buckets(torch, "cuda:0", steps=steps, batches=batches, shapes=shapes, log=log)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"workload": "one training step (bf16, V=500, Adam) per real bucket of configs/data.json (after /2), lengths per latex_ocr_amd.synthetic.bucket_lengths", "rows": rows},
          open(os.path.join(ROOT, "gpurun_out", "%s_buckets.json" % tag), "w"), indent=1)
