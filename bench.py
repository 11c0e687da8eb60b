#!/usr/bin/env python
"""Headline benchmark: formula-images/sec of one full training step (encoder + attention
decoder forward, loss, BPTT, Adam, weight re-pack [+ gradient all-reduce]) on synthetic
128x512 crops, batch 64 per GPU, vocab 500, bf16 storage / f32 accumulate
(BASELINE.json configs[2]; metric quoted at 1/2/4/8 MI355X, weak scaling).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GF_TRAIN_PER_IMG = 55.925e9      # conv fwd+dgrad+wgrad FLOPs per 128x512 image
MFMA_BF16_PEAK = 2.5e15          # MI355X_MICROARCH.md: dense bf16 MFMA peak


def conv_layers(B, H, W, C=512):
    """(name, M, N, K) of the implicit GEMMs of conv2..conv6 forward at this input size."""
    c = lambda n: -(-n // 2)
    H1, W1 = c(H), c(W); H2, W2 = c(H1), c(W1); H4 = c(H2); W5 = c(W2)
    return [("conv2", B * H1 * W1, 128, 9 * 64), ("conv3", B * H2 * W2, 256, 9 * 128), ("conv4", B * H2 * W2, 256, 9 * 256),
            ("conv5", B * H4 * W2, C, 9 * 256), ("conv6", B * (H4 - 2) * (W5 - 2), C, 9 * C)]


def cpu_baseline(seconds_budget=15.0):
    """The oracle restatement of the reference trainer (kind "port": TF-1.12 cannot run here),
    timed on the host cores on a bounded sample of the same workload: batch 2 of 128x512,
    vocab 500, formula lengths U{30..100}."""
    import torch
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    from oracle import ref_model as R
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))           # torch intra-op threads actually used (more only adds spin-wait on big hosts)
    torch.set_num_threads(cores)
    V, B = 500, 2
    imgs, forms = synthetic.make_set(B, 128, 512, V, 30, 101, seed=99)
    img = torch.from_numpy(pad_batch_images(imgs))
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    f, l = torch.from_numpy(f), torch.from_numpy(l)
    P = R.init_params(V, 0)
    opt = R.AdamTF(P)
    tw = time.time()
    R.train_step(P, opt, img, f, l, 1e-4)          # warm-up (also sizes the sample: never more than ~budget seconds)
    tw = time.time() - tw
    max_steps = int(max(1, min(10, seconds_budget / max(tw, 1e-3))))
    t0 = time.time(); n = 0
    while n < max_steps:
        R.train_step(P, opt, img, f, l, 1e-4); n += 1
        if time.time() - t0 > seconds_budget:
            break
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "oracle/ref_model.py train_step (torch-CPU f32), %d steps of batch %d, 128x512, V=500, T=%d"
                      % (n, B, f.shape[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=500)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline timing only (for profiler runs): no roofline micro-timings, no secondary keys")
    args = ap.parse_args()
    if args.dtype != "bf16":
        ap.error("bench.py measures the bf16 path (the metric's dtype); f32 is the parity mode exercised by tests/")

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1 or os.environ.get("LXO_FORCE_DIST") == "1":
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        from latex_ocr_amd.dist import DataParallel
        dist = DataParallel(device=dev)

    from latex_ocr_amd import synthetic
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas

    B, H, W, V = args.batch, args.height, args.width, args.vocab
    imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234 + rank)
    img = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    f_d = torch.from_numpy(f).to(dev)
    T = int(f.shape[1])
    eng = Engine(V, dtype=args.dtype, device=dev, seed=0)

    def step():
        return eng.train_step(img, f_d, l, 1e-3, dist=dist, sync_loss=False)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce_max(tt)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)

    out = None
    if rank == 0 and args.no_extras:
        print(json.dumps({"metric": "formula-images/sec training step (batch 64, 128x512)", "value": round(value, 2), "unit": "img/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3)}), flush=True)
    elif rank == 0:
        # ---- roofline of the dominant kernel family: the implicit-GEMM conv (MFMA bound) ----
        flops, secs, per = eng.time_conv_gemms(B, H, W)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "conv_nt_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # dominant kernel: conv_halo2wg_kernel (8x32-pixel halo tiles x 128 / 64 channels, two workgroups per CU; all ten conv launches)
        dom = [x for x in per if x["kernel"] == "conv_halo2wg_kernel"] or per
        dflops, dsecs = sum(x["flops"] for x in dom), sum(x["us"] for x in dom) * 1e-6
        for x in per:
            x.pop("flops", None)
        roof = {"kernel": "%s (bf16 implicit-GEMM 3x3 conv, halo tiles): %d of the %d conv forward/dgrad launches of a step"
                          % (dom[0]["kernel"], len(dom), len(per)),
                "bound": "mfma", "achieved": round(dflops / dsecs / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(dflops / dsecs / MFMA_BF16_PEAK, 4), "traffic": traffic,
                "flops_per_launch": dflops / len(dom), "avg_launch_us": round(dsecs / len(dom) * 1e6, 1),
                "all_conv_launches": {"achieved": round(flops / secs / 1e12, 2), "frac": round(flops / secs / MFMA_BF16_PEAK, 4),
                                      "launches": per}}
        c8 = lambda n: -(-(-(-(-(-n // 2)) // 2)) // 2)
        Rr = (c8(H) - 2) * (c8(W) - 2)
        nbytes, asec = eng.time_attention(B, Rr)
        roof_att = {"kernel": "attn_fwd_part_kernel + attn_fwd_combine_kernel (one decoder step, B samples)", "bound": "hbm",
                    "achieved": round(nbytes / asec / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / asec / 8e12, 4),
                    "traffic": None, "bytes_per_launch": nbytes, "avg_launch_us": round(asec * 1e6, 2)}
        out = {
            "metric": "formula-images/sec training step (batch 64, 128x512)", "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[2]: full encoder+attention+decoder training step, batch %d/GPU, %dx%d, vocab %d, "
                                   "T=%d (lengths U{30..100}), Adam" % (B, H, W, V, T),
                       "global_batch": B * world, "parallelism": "dp%d" % world},
            "conv_roofline_fraction_e2e": round(GF_TRAIN_PER_IMG * (H * W / (128.0 * 512.0)) * value / world / MFMA_BF16_PEAK, 4),
        }
        out["roofline"] = roof
        out["roofline_attention"] = roof_att
        if world == 1:
            # secondary, NOT the headline: the opt-in extension that runs each decoder step only for the samples still
            # inside their formula (same loss and gradients; the reference and `value` above run every padded step)
            eng.skip_padded = True
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / args.steps
            eng.skip_padded = False
            out["extension_skip_padded_steps"] = {"value": round(B / dts, 2), "unit": "img/s", "ms_per_step": round(dts * 1e3, 3),
                                                  "note": "not the headline metric: padded (sample, step) pairs skipped, batch sorted by length"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        import torch.distributed as td
        dist.barrier()                   # ranks > 0 wait for rank 0's roofline micro-timings before tearing down
        td.destroy_process_group()


if __name__ == "__main__":
    main()
