#!/usr/bin/env python
"""Headline benchmark: formula-images/sec of one full training step (encoder + attention
decoder forward, loss, BPTT, Adam, weight re-pack [+ gradient all-reduce]) on synthetic
128x512 crops, batch 64 per GPU, vocab 500, bf16 storage / f32 accumulate
(BASELINE.json configs[2]; metric quoted at 1/2/4/8 MI355X, weak scaling).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torchrun environment: this process re-launches itself as N ranks
      (python -m torch.distributed.run --nproc-per-node N ..., one rank per GPU, RCCL).
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
      (what the driver does) works as before: RANK / LOCAL_RANK / WORLD_SIZE come from the env.

Rank 0 prints ONE JSON line.  After the timed region rank 0 runs ONE more, instrumented step (HIP events
around every conv / attention launch inside the step, lxo_timing_*) from which the roofline entries are computed,
then the secondary measurements (none of them touches `value`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GF_TRAIN_PER_IMG = 55.925e9      # conv fwd+dgrad+wgrad FLOPs per 128x512 image
MFMA_BF16_PEAK = 2.5e15          # MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK = 8.0e12                # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
METRIC = "formula-images/sec training step (batch 64, 128x512)"


def cpu_baseline(seconds_budget=20.0):
    """The oracle restatement of the reference trainer (kind "port": TF-1.12 cannot run here),
    timed on the host cores on a bounded sample of the same workload: batch 4 of 128x512 (B = 4 .. 8),
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
    V, B = 500, 4
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """--gpus N > 1 outside torchrun: become the launcher of N ranks of this same script."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def pin_rank_to_numa(local_rank, world, torch):
    """One host thread per GPU drives ~3 ms of launch work per step: give every rank its own cores, the ones local to ITS GPU's
    NUMA node when sysfs tells (pci local_cpulist), else an even slice of the cores this process may use.  Best effort."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except Exception:
        return None
    local = None
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        local = [c for c in avail if c in cpus]
    except Exception:
        local = None
    pool = local if local else avail
    per = max(1, len(pool) // max(1, world))
    # ranks sharing a NUMA node (or the whole pool) take consecutive slices by local rank
    mine = pool[(local_rank * per) % len(pool):][:per] or pool
    try:
        os.sched_setaffinity(0, mine)
    except Exception:
        return None
    return {"cores": len(mine), "numa_local": bool(local)}


# ------------------------------------------------------------------ measurement helpers (rank 0, after the timed region) ----
def timing_records(lib):
    import ctypes
    out = []
    for i in range(lib.lxo_timing_count()):
        fam, nm = ctypes.c_char_p(), ctypes.c_char_p()
        work, ms = ctypes.c_double(), ctypes.c_float()
        if lib.lxo_timing_get(i, ctypes.byref(fam), ctypes.byref(nm), ctypes.byref(work), ctypes.byref(ms)) == 0:
            out.append((fam.value.decode(), nm.value.decode(), work.value, ms.value * 1e-3))
    return out


def instrumented_step(eng, img, f_d, l, torch):
    """One more real training step with per-launch HIP events (lxo_timing_*) and per-call events around the ABI calls."""
    lib = eng.lib
    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks = []

    def mark(name):
        e = ev(); e.record(); marks.append((name, e))
    lib.lxo_timing_enable(1)
    torch.cuda.synchronize()
    mark("start")
    eng.forward(img, f_d, phase_hook=mark)
    mark("decoder_fwd")
    eng.loss(l, 1.0 / float(np.asarray(l).sum()))
    mark("loss")
    eng.backward(phase_hook=mark)
    mark("encoder_bwd")
    eng.optimizer_step(1e-3)
    mark("optimizer+pack")
    torch.cuda.synchronize()
    lib.lxo_timing_enable(0)
    phases = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        phases[n1] = phases.get(n1, 0.0) + e0.elapsed_time(e1)
    return timing_records(lib), {k: round(v, 3) for k, v in phases.items()}


def roofline_from_records(recs, family, label, bound, peak, unit):
    sel = [r for r in recs if r[0] in family]
    if not sel:
        return None
    work = sum(r[2] for r in sel); secs = sum(r[3] for r in sel)
    scale = 1e12 if unit == "TFLOP/s" else 1e9
    per = {}
    for fam, nm, w, s in sel:
        k = fam + ":" + nm
        a = per.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += w; a[2] += s
    return {"kernel": label, "bound": bound, "achieved": round(work / secs / scale, 2), "peak": peak / scale, "unit": unit,
            "frac": round(work / secs / peak, 4), "traffic": None, "launches": len(sel),
            "work_per_launch": work / len(sel), "avg_launch_us": round(secs / len(sel) * 1e6, 2),
            "source": "HIP events around each launch inside one real training step (lxo_timing_*); that step keeps every launch on ONE stream -- in the timed steps the weight-gradient launches run on a second stream beside the data-gradient launches (LXO_ENC_OVERLAP, default on), which would put two kernels inside one bracket",
            "per_launch": {k: {"n": v[0], "us": round(v[2] / v[0] * 1e6, 2), "rate": round(v[1] / v[2] / scale, 1)} for k, v in sorted(per.items())}}


def chain_phases(eng, torch, backward=False):
    """Per-phase time of one step of the persistent XCD-local decoder chain (csrc/xdec.hip; backward=True: its BPTT twin) from the
    in-kernel 100 MHz timestamps (lxo_xdec_debug / lxo_xdec_debug_bwd): mean over the 256 workgroups and the steps of one more
    decoder forward (backward) on the bound shape.  None when the chain does not run for this engine."""
    import ctypes
    from latex_ocr_amd.engine import _p
    used, err = eng.chain_status(backward=backward)
    if not used or err:
        return None
    T = int(eng.shape.T)
    buf = torch.zeros(256 * T * 16, dtype=torch.int64, device=eng.device)
    hook = eng.lib.lxo_xdec_debug_bwd if backward else eng.lib.lxo_xdec_debug
    hook.argtypes = [ctypes.c_void_p]
    hook(ctypes.c_void_p(buf.data_ptr()))
    try:
        if backward:
            eng._ck(eng.lib.lxo_decoder_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng.grads), eng._stream()), "bwd")
        else:
            eng._ck(eng.lib.lxo_decoder_train_fwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), eng._stream()), "fwd")
        torch.cuda.synchronize()
    finally:
        hook(ctypes.c_void_p(0))
    s = buf.cpu().numpy().reshape(256, T, 16).astype(np.float64) * 0.01            # us
    d = s[:, 2:, 1:9] - s[:, 2:, 0:8]
    # round 6: the chunk partials are polled hand-over words, no barrier behind the stream phase; the wait for them sits at the head of the NEXT phase
    # (P4 / Q3) and ends at stamp 12 (0 where the build keeps the barrier)
    poll = s[:, 2:, 12] - s[:, 2:, 4 if backward else 6]
    poll_wait = float(poll[poll > 0].mean()) if (poll > 0).any() else 0.0
    names = (["Q1_dh_dctx", "barrier1", "Q2_attention_stream", "barrier2", "Q3_lstm_bwd", "barrier3", "Q4_carries", "barrier4"] if backward else
             ["P1_lstm", "barrier1", "P2_att_h", "barrier2", "P3_attention_stream", "barrier3", "P4_merge_o", "barrier4"])
    out = {n: round(float(d[:, :, i].mean()), 3) for i, n in enumerate(names)}
    out["step_us"] = round(float((s[:, 2:, 8] - s[:, 2:, 0]).mean()), 3)
    out["partials_poll_wait"] = round(poll_wait, 3)
    return out


def pmc_traffic_inrun(patterns, timeout=200):
    """HBM traffic per launch of the named kernels, measured NOW: this script re-runs itself for two short passes under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes) and
    reads the counters back: bytes = 2 x FETCH_SIZE KB (gfx950 correction) + WRITE_SIZE KB.  None if rocprofv3 is not usable."""
    import shutil, sqlite3, tempfile, glob
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    res = {}
    tmp = tempfile.mkdtemp(prefix="lxo_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-extras", "--no-secondary"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
            dbs = glob.glob(os.path.join(d, "*", "*_results.db")) + glob.glob(os.path.join(d, "*_results.db"))
            if r.returncode != 0 or not dbs:
                return None
            db = sqlite3.connect(dbs[0])
            for pat in patterns:
                n, v = db.execute("select count(*), avg(value) from counters_collection where kernel_name like ? and counter_name = ?",
                                  ("%" + pat + "%", counter)).fetchone()
                if n and v is not None:
                    res.setdefault(pat, {"dispatches": n})[counter] = float(v)
            db.close()
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for pat, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[pat] = {"dispatches": v["dispatches"], "fetch_bytes_per_launch": 2.0 * v["FETCH_SIZE"] * 1024.0, "write_bytes_per_launch": v["WRITE_SIZE"] * 1024.0,
                        "hbm_bytes_per_launch": 2.0 * v["FETCH_SIZE"] * 1024.0 + v["WRITE_SIZE"] * 1024.0}
    return out or None


def count_set(n, seed, H, W):
    """Crops whose ink density tells the formula length (tests/test_gpu_benchcfg.py): ~250 Adam steps teach the model a
    per-class count-down to END, which is what a decode measurement with early exit needs."""
    classes = [(0.0, 2), (0.08, 4), (0.3, 9), (0.6, 5)]
    rng = np.random.Generator(np.random.PCG64(seed))
    imgs, forms = [], []
    for _ in range(n):
        ink, L = classes[int(rng.integers(0, len(classes)))]
        im = np.full((H, W, 1), 255, np.uint8)
        m = rng.random((H, W, 1)) < ink
        vals = rng.integers(0, 128, size=(H, W, 1)).astype(np.uint8)
        im[m] = vals[m]
        imgs.append(im); forms.append([7] * L)
    return imgs, forms


def secondary(args, eng_train, torch, dev, B, H, W, V):
    """configs[1] (encoder only, B=32), the worst-case T=151 step, configs[4] decode tokens/s.  Separate engines; never `value`."""
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    out = {}

    def timed(fn, n, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # ---- configs[1]: encoder-only conv kernels, batch 32 ----
    eng = Engine(V, dtype="bf16", device=dev, seed=0)
    imgs, forms = synthetic.make_set(32, H, W, V, 3, 5, seed=7)
    img32 = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    dt = timed(lambda: eng._encode_only(img32, 1), 20)
    out["config2_encoder_only_b32"] = {"fwd_ms": round(dt * 1e3, 3), "fwd_img_per_s": round(32 / dt, 1),
                                       "conv_fwd_tflops": round(18.667e9 * (H * W / 65536.0) * 32 / dt / 1e12, 1)}
    # ---- worst case: every formula at max_length_formula = 150 (T = 151) ----
    imgs, forms = synthetic.make_set(B, H, W, V, 150, 151, seed=8)
    imgT = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    fT, lT = pad_batch_formulas(forms, V - 2, V - 1)
    fT_d = torch.from_numpy(fT).to(dev)
    dt = timed(lambda: eng.train_step(imgT, fT_d, lT, 1e-3, sync_loss=False), 10)
    out["worst_case_T151"] = {"ms_per_step": round(dt * 1e3, 3), "img_per_s": round(B / dt, 1), "T": int(fT.shape[1])}
    # ---- configs[4]: decode.  (a) at the step bound (random weights never emit END: 152 steps), (b) with weights trained
    #      to END by the count-down recipe (early exit) ----
    dec_img = imgT
    # id_end = -1 can never be emitted: every row runs to the bound (max_iter + 1 = 152 steps), as an untrained model does
    ns = int(eng.greedy_decode(dec_img, -1, max_iter=151).shape[1])
    dt = timed(lambda: eng.greedy_decode(dec_img, -1, max_iter=151), 3, warm=1)
    dte = timed(lambda: eng._encode_only(dec_img, 1), 5, warm=1)           # the encoder of the same batch alone: what the per-step figures below exclude
    used, err = eng.chain_status()
    out["decode_greedy_bound"] = {"ms_per_batch": round(dt * 1e3, 2), "steps": ns, "us_per_step": round(dt * 1e6 / ns, 1),
                                  "tokens_per_s": round(B * ns / dt, 0), "batch": B, "encoder_ms": round(dte * 1e3, 3),
                                  "loop_us_per_step": round((dt - dte) * 1e6 / ns, 1), "persistent_chain": bool(used and not err),
                                  "note": "us_per_step = the whole call (encoder + decode loop) / steps; loop_us_per_step leaves the encoder out; the loop runs as the persistent decode chain (xdec_dec_kernel, 16 steps per launch)"}
    ns = int(eng.beam_decode(dec_img, -1, 5, max_iter=151).shape[1])
    dt = timed(lambda: eng.beam_decode(dec_img, -1, 5, max_iter=151), 2, warm=1)
    out["decode_beam5_bound"] = {"ms_per_batch": round(dt * 1e3, 2), "steps": ns, "us_per_step": round(dt * 1e6 / ns, 1),
                                 "tokens_per_s": round(B * ns / dt, 0), "batch": B, "beam": 5, "loop_us_per_step": round((dt - dte) * 1e6 / ns, 1),
                                 "note": "tokens = emitted positions of the best hypothesis per image (B x steps); the encoder is inside the timing (us_per_step), outside loop_us_per_step"}
    engc = Engine(V, dtype="bf16", device=dev, seed=0)
    for step in range(260):
        ci, cf = count_set(16, 100 + step, H, W)
        f, l = pad_batch_formulas(cf, V - 2, V - 1)
        engc.train_step(pad_batch_images(ci), f, l, 1e-3, sync_loss=False)
    ci, cf = count_set(B, 5, H, W)
    cimg = torch.from_numpy(pad_batch_images(ci)).to(dev)
    ids = engc.greedy_decode(cimg, V - 1, max_iter=151)
    ntok = int(sum(min(len(x) + 1, ids.shape[1]) for x in cf))
    dt = timed(lambda: engc.greedy_decode(cimg, V - 1, max_iter=151), 5, warm=1)
    out["decode_greedy_trained_to_end"] = {"ms_per_batch": round(dt * 1e3, 2), "steps": int(ids.shape[1]), "tokens_per_s": round(ntok / dt, 0),
                                           "batch": B, "note": "weights: 260 Adam steps on the ink-density count-down set; tokens = formula tokens + END"}
    bids = engc.beam_decode(cimg, V - 1, 5, max_iter=151)
    dt = timed(lambda: engc.beam_decode(cimg, V - 1, 5, max_iter=151), 5, warm=1)
    out["decode_beam5_trained_to_end"] = {"ms_per_batch": round(dt * 1e3, 2), "steps": int(bids.shape[1]), "tokens_per_s": round(ntok / dt, 0),
                                          "batch": B, "beam": 5}
    # ---- bf16 deterministic mode (lxo_shape.deterministic: every reduction ordered, bit-identical runs): what it costs on the headline workload ----
    try:
        engd = Engine(V, dtype="bf16", device=dev, seed=0, deterministic=True)
        imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
        imgd = torch.from_numpy(pad_batch_images(imgs)).to(dev)
        fd, ld = pad_batch_formulas(forms, V - 2, V - 1)
        fd_d = torch.from_numpy(fd).to(dev)
        dtd = timed(lambda: engd.train_step(imgd, fd_d, ld, 1e-3, sync_loss=False), 10, warm=3)
        out["deterministic_bf16"] = {"ms_per_step": round(dtd * 1e3, 3), "img_per_s": round(B / dtd, 1), "chains": bool(engd.chain_used and engd.chain_used_bwd),
                                     "note": "the headline workload with Engine(deterministic=True): conv and dense weight gradients through per-range slabs + an ordered pass, bias sums / d_beta / loss / conv1 through ordered per-workgroup slots (no float atomics: two runs agree bit for bit, tests/test_gpu_determinism.py); tools/det_ab.py alternates the two modes in one process (profiles/r05_det_ab_two_streams.txt: +0.7 %; with one stream +1.9 %)"}
        del engd
    except Exception as e:
        out["deterministic_bf16"] = {"error": repr(e)}
    # ---- data parallel at world size 1 on this GPU: the N > 1 code path (token-count all-reduce on its stream, gradient buckets on the side
    #      stream through lxo_allreduce_bucket on a one-rank RCCL communicator, chain guard behind the exchange) and what it costs ----
    try:
        import torch.distributed as td
        if not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            td.init_process_group(backend="gloo", rank=0, world_size=1)
            from latex_ocr_amd.dist import DataParallel
            dp = DataParallel(device=dev)
            dp.time_finish = True
            engp = Engine(V, dtype="bf16", device=dev, seed=0)
            imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
            imgp = torch.from_numpy(pad_batch_images(imgs)).to(dev)
            fp, lp = pad_batch_formulas(forms, V - 2, V - 1)
            fp_d = torch.from_numpy(fp).to(dev)
            dtp1 = timed(lambda: engp.train_step(imgp, fp_d, lp, 1e-3, dist=dp, sync_loss=False), 20, warm=5)
            ex = dp.exposed_allreduce_ms()
            out["data_parallel_world1"] = {"ms_per_step": round(dtp1 * 1e3, 3), "img_per_s": round(B / dtp1, 1),
                                           "rccl_ranks_seen": dp.lxo.ranks_seen if dp.lxo is not None else None,
                                           "exposed_allreduce_ms_per_step": None if ex is None else round(ex, 3),
                                           "chains": bool(engp.chain_used and engp.chain_used_bwd), "chain_failures": engp.chain_failures,
                                           "bucket_order": "host-ordered" if dp.host_ordered else "stream-ordered",
                                           "note": "ONE rank: the data-parallel step as the driver's N > 1 runs execute it, on a one-rank RCCL communicator behind the C ABI; NO scaling curve has been measured by this repository (one GPU per gpurun box)"}
            dp.close()
            td.destroy_process_group()
            del engp
    except Exception as e:
        out["data_parallel_world1"] = {"error": repr(e)}
    # ---- the optional row-BiLSTM encoder (north_star names it; not in the reference; off in the headline) ----
    engr = Engine(V, dtype="bf16", device=dev, seed=0, dims=dict(row_bilstm=True))
    imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
    imgr = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    fr, lr_ = pad_batch_formulas(forms, V - 2, V - 1)
    fr_d = torch.from_numpy(fr).to(dev)
    dt = timed(lambda: engr.train_step(imgr, fr_d, lr_, 1e-3, sync_loss=False), 5)
    out["encoder_row_bilstm"] = {"ms_per_step": round(dt * 1e3, 3), "img_per_s": round(B / dt, 1),
                                 "note": "same workload with the optional row encoder (bidirectional LSTM, 256 units per direction, over the 14 rows x 62 positions of every feature map) between conv6 and the decoder"}
    del engr
    # ---- the input pipeline in the loop (the reference feeds fresh host arrays every step, img2seq.py:160-169): Prefetcher =
    #      background pad_batch_images / pad_batch_formulas into pinned buffers + H2D on the copy stream, `depth` batches ahead ----
    from latex_ocr_amd.pipeline import Prefetcher
    from latex_ocr_amd.model.utils.data_generator import ListDataset
    pool_i, pool_f = synthetic.make_set(256, H, W, V, 30, 101, seed=4321)
    nsteps = 50
    idx = [(s * B + j) % 256 for s in range(nsteps + 4) for j in range(B)]
    ds = ListDataset([pool_i[i] for i in idx], [pool_f[i] for i in idx])
    t_host = time.perf_counter()
    for s in range(4):                                   # host cost of one batch on ONE thread: pad images + formulas (+ pinning is in the loader)
        pad_batch_images(ds.images[s * B:(s + 1) * B]); pad_batch_formulas(ds.formulas[s * B:(s + 1) * B], V - 2, V - 1)
    t_host = (time.perf_counter() - t_host) / 4
    feed = iter(Prefetcher(ds, B, V - 2, V - 1, device=dev, depth=3))
    for _ in range(4):
        b = next(feed); eng_train.train_step(b.img, b.formula, b.lengths, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for b in feed:
        eng_train.train_step(b.img, b.formula, b.lengths, 1e-3, sync_loss=False); n += 1
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / max(n, 1)
    try:
        rows = real_buckets(torch, dev, V=V, batches=(3, 20, 64), steps=4, warm=2)
        keep = ("H", "W", "B", "chain_batch", "T", "regions", "ms_per_step", "img_per_s", "us_per_image", "chains", "conv_fwd_dgrad_frac", "conv_wgrad_frac",
                "chain_fwd_us_per_step", "chain_bwd_us_per_step", "error")
        out["real_buckets"] = {"rows": [{k: r[k] for k in keep if k in r} for r in rows],
                               "note": "one training step (bf16, V=%d, Adam) on each of the reference's 21 image sizes (configs/data.json:22-28 after the build-time /2 downsample) at the reference's training batch 3 (configs/training.json:6; chain batch 8), its bucket / evaluation batch 20 (data_generator.py:41, evaluate_txt.py:42; filled up to a chain batch of 32 with dead rows) and at 64; formula lengths per latex_ocr_amd.synthetic.bucket_lengths; conv fractions = algorithmic FLOPs of THAT shape / HIP-event time of the launches / 2.5 PFLOP/s; chain us per step = whole persistent launch / T; 4 timed steps per row (tools/real_buckets.py takes 8; profiles/r06_buckets.json)" % V}
    except Exception as e:
        out["real_buckets"] = {"error": repr(e)}
    out["pipeline_fed"] = {"img_per_s": round(B / dtp, 1), "ms_per_step": round(dtp * 1e3, 3), "steps": n, "host_pad_ms_per_batch": round(t_host * 1e3, 2),
                           "h2d_bytes_per_batch": int(B * H * W + B * 101 * 4),
                           "note": "fresh host lists every step through latex_ocr_amd.pipeline.Prefetcher (depth 3): padding on a background thread, pinned staging, copy stream; T varies per batch (batch-max + 1)"}
    return out


def real_buckets(torch, dev, V=500, batches=(3, 20, 64), steps=8, warm=3, shapes=None, log=None):
    """SURVEY 8(f)2 / VERDICT r5 #1: the training step on the reference's REAL image sizes (latex_ocr_amd.synthetic.REAL_BUCKETS =
    configs/data.json:22-28 after the /2 downsample) at the reference's bucket / evaluation batch (20: data_generator.py:41,
    evaluate_txt.py:42) and at the benchmark's 64.  Per bucket: ms per step, img/s, us per image, the conv fwd+dgrad and wgrad
    fractions of the bf16 MFMA peak (HIP events around each launch of one instrumented step, algorithmic FLOPs of THAT shape), the
    decoder chains' us per step.  Never `value`."""
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.model.utils.image import pad_batch_images, encoder_out_hw
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    rows = []
    for (H, W) in (shapes or synthetic.REAL_BUCKETS):
        lo, hi = synthetic.bucket_lengths(W)
        for B in batches:
            try:
                eng = Engine(V, dtype="bf16", device=dev, seed=0)
                imgs, forms = synthetic.make_set(B, H, W, V, lo, hi, seed=1234 + H + W)
                img = torch.from_numpy(pad_batch_images(imgs)).to(dev)
                f, l = pad_batch_formulas(forms, V - 2, V - 1)
                f_d = torch.from_numpy(f).to(dev)
                for _ in range(warm):
                    eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                recs, phases = instrumented_step(eng, img, f_d, l, torch)
                T = int(f.shape[1])
                Hp, Wp = encoder_out_hw(H, W)

                def fam(names):
                    sel = [r for r in recs if r[0] in names]
                    w = sum(r[2] for r in sel); s_ = sum(r[3] for r in sel)
                    return (round(w / s_ / MFMA_BF16_PEAK, 4), round(s_ * 1e3, 4)) if sel and s_ > 0 else (None, None)
                cf, cms = fam(("conv_fwd", "conv_dgrad"))
                wf, wms = fam(("conv_wgrad",))
                xf = [r for r in recs if r[0] == "xdec_fwd"]; xb = [r for r in recs if r[0] == "xdec_bwd"]
                row = {"H": H, "W": W, "B": B, "chain_batch": int(eng.shape.B), "T": T, "regions": Hp * Wp,
                       "ms_per_step": round(dt * 1e3, 3), "img_per_s": round(B / dt, 1), "us_per_image": round(dt * 1e6 / B, 1),
                       "chains": bool(eng.chain_used and eng.chain_used_bwd),
                       "conv_fwd_dgrad_frac": cf, "conv_fwd_dgrad_ms": cms, "conv_wgrad_frac": wf, "conv_wgrad_ms": wms,
                       "chain_fwd_us_per_step": round(xf[0][3] * 1e6 / T, 2) if xf else None,
                       "chain_bwd_us_per_step": round(xb[0][3] * 1e6 / T, 2) if xb else None,
                       "ms_by_phase_one_stream": phases}
                del eng, img, f_d
            except Exception as e:
                row = {"H": H, "W": W, "B": B, "error": repr(e)}
            rows.append(row)
            if log:
                log(row)
            torch.cuda.empty_cache()
    return rows


def cpu_baselines_secondary(eng_train, torch, dev, V):
    """The CPU side of the non-training configurations, bounded samples, never `value`:
    beam-5 decode tokens/s of the oracle (beam_search_decoder_cell.py:123-187 restated) and configs[0] end to end
    (1 epoch of 5 Adam steps at batch 20 on 100 crops of 32x128, V = 50, then greedy decode of all 100 crops) on both sides."""
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.model.utils.general import minibatches
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    from oracle import ref_model as R
    out = {}
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    # ---- beam 5 on the CPU: batch 2, 128x512, V = 500, 20 steps (END never emitted: the untrained weights of the GPU 'bound' run) ----
    imgs, _ = synthetic.make_set(2, 128, 512, V, 3, 5, seed=8)
    img = torch.from_numpy(pad_batch_images(imgs))
    P = R.init_params(V, 0)
    t0 = time.perf_counter()
    ids, _ = R.beam_decode(P, img, -1, 5, max_iter=19)
    dt = time.perf_counter() - t0
    out["cpu_baseline_decode_beam5"] = {"value": round(2 * ids.shape[1] / dt, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
                                        "sample": "oracle/ref_model.py beam_decode (torch-CPU f32), batch 2, beam 5, 128x512, V=%d, %d steps incl. the encoder; tokens = B x steps as in secondary.decode_beam5_bound" % (V, ids.shape[1])}
    # ---- configs[0] end to end on both sides ----
    Vs = 50
    imgs, forms = synthetic.config1()
    def run_cpu():
        P = R.init_params(Vs, 0); opt = R.AdamTF(P)
        for bi, bf in minibatches(zip(imgs, forms), 20):
            f, l = pad_batch_formulas(bf, Vs - 2, Vs - 1)
            R.train_step(P, opt, torch.from_numpy(pad_batch_images(bi)), torch.from_numpy(f), torch.from_numpy(l), 1e-3)
        return R.greedy_decode(P, torch.from_numpy(pad_batch_images(imgs)), Vs - 1, max_iter=151).shape[1]
    def run_gpu(dtype):
        eng = Engine(Vs, dtype=dtype, device=dev, seed=0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for bi, bf in minibatches(zip(imgs, forms), 20):
            f, l = pad_batch_formulas(bf, Vs - 2, Vs - 1)
            eng.train_step(pad_batch_images(bi), f, l, 1e-3)
        n = eng.greedy_decode(pad_batch_images(imgs), Vs - 1, max_iter=151).shape[1]
        torch.cuda.synchronize()
        return time.perf_counter() - t, n
    t0 = time.perf_counter(); ncpu = run_cpu(); tcpu = time.perf_counter() - t0
    run_gpu("bf16")                                      # warm-up (kernel attributes, allocator)
    tg, ng = run_gpu("bf16")
    out["config1_end_to_end"] = {"gpu_seconds": round(tg, 4), "cpu_seconds": round(tcpu, 3), "cpu_cores": cores, "decode_steps": [int(ng), int(ncpu)],
                                 "sample": "configs[0]: 100 synthetic 32x128 crops, V=50: 1 epoch (5 Adam steps, batch 20, host padding and uploads included) + greedy decode of the 100 crops to the bound; GPU = bf16 engine, CPU = oracle port (torch-CPU f32)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--vocab", type=int, default=500)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline timing only (profiler runs): no instrumented step, no secondary keys")
    ap.add_argument("--no-secondary", action="store_true", help="skip configs[1] / T=151 / decode measurements")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two in-run rocprofv3 --pmc passes (HBM traffic of the roofline kernels)")
    ap.add_argument("--sim", action="store_true", help="TEST ONLY: CPU tensors, gloo, the hipsim build of the kernels, tiny shapes")
    args = ap.parse_args()
    if args.dtype != "bf16" and not args.sim:
        ap.error("bench.py measures the bf16 path (the metric's dtype); f32 is the parity mode exercised by tests/")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas

    dist = None
    if args.sim:
        # the N > 1 path on a GPU-less box: same driver code, gloo instead of RCCL, kernels interpreted by tests/hipsim
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from simlib import SIM_SO, build_sim
        from latex_ocr_amd import _abi
        if rank == 0:
            build_sim()
        dev = "cpu"
        sync = lambda: None
        if world > 1:
            import torch.distributed as td
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group(backend="gloo", rank=rank, world_size=world)
            td.barrier()
            from latex_ocr_amd.dist import DataParallel
            dist = DataParallel(device="cpu")
        small = dict(C=128, E=128, U=128, O=128, D=16)
        B, H, W, V = 2, 32, 48, 11
        eng = Engine(V, dims=small, dtype="f32", device="cpu", seed=0, lib=_abi.bind(ctypes.CDLL(SIM_SO)))
        imgs, forms = synthetic.make_set(B, H, W, V, 2, 5, seed=1234 + rank)
    else:
        torch.cuda.set_device(local_rank)
        dev = "cuda:%d" % local_rank
        sync = torch.cuda.synchronize
        pinned = pin_rank_to_numa(local_rank, world, torch) if world > 1 else None
        if world > 1 or os.environ.get("LXO_FORCE_DIST") == "1":
            import torch.distributed as td
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # control plane = a gloo group (the 128-byte RCCL id, host barriers, the final gathers); data plane = RCCL over xGMI through
            # liblxo's C ABI (lxo_comm_init / lxo_allreduce_bucket); LXO_DP_COMM=torch puts both on a torch.distributed nccl group instead
            if os.environ.get("LXO_DP_COMM", "abi") == "torch":
                td.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))
            else:
                td.init_process_group(backend="gloo", rank=rank, world_size=world)
            from latex_ocr_amd.dist import DataParallel
            dist = DataParallel(device=dev)
            dist.time_finish = True                              # event pair around the wait for the gradient buckets
            if os.environ.get("LXO_DP_BF16") == "1":
                dist.grad_dtype = torch.bfloat16                 # opt-in: gradients cross xGMI as bf16 (17 MB instead of 34.5 MB)
        B, H, W, V = args.batch, args.height, args.width, args.vocab
        eng = Engine(V, dtype=args.dtype, device=dev, seed=0)
        imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234 + rank)
    img = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    f_d = torch.from_numpy(f).to(dev)
    T = int(f.shape[1])

    def step():
        return eng.train_step(img, f_d, l, 1e-3, dist=dist, sync_loss=False)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    sync()
    if dist is not None and getattr(dist, "exposed_ms", None) is not None:
        dist.exposed_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    dp_info = None
    if dist is not None:
        # per-rank step time and exposed all-reduce wait, gathered before the MAX that defines `value`
        mine = torch.zeros(world, 6, dtype=torch.float64, device=dev)
        ex = dist.exposed_allreduce_ms() if not args.sim else None
        if getattr(eng, "_health_ring", None) is not None:
            eng._chain_health_poll(wait=True)                    # the device is idle (sync above): every posted step's chain error words are in
        mine[rank, 0] = dt / args.steps * 1e3
        mine[rank, 1] = -1.0 if ex is None else ex
        mine[rank, 2] = eng.chain_failures
        mine[rank, 3] = getattr(eng, "dropped_steps", 0)
        mine[rank, 4] = 1.0 if eng.step_kernels != int(os.environ.get("LXO_STEP_KERNELS", "0")) else 0.0     # this rank left the persistent chains for the launch-per-step kernels
        mine[rank, 5] = 1.0 if (eng.chain_used and eng.chain_used_bwd) else 0.0
        dist.all_reduce(mine)
        dp_info = {"per_rank_ms_per_step": [round(float(x), 3) for x in mine[:, 0].tolist()],
                   "exposed_allreduce_ms_per_step": [round(float(x), 3) for x in mine[:, 1].tolist()],
                   # health of the persistent decoder chains next to RCCL's kernels, per rank: a chain that does not assemble drops its step on
                   # EVERY rank (lxo_chain_guard) and moves that rank to the launch-per-step kernels -- a silent 8-GPU slow path must show here
                   "chains_ran": [bool(x) for x in mine[:, 5].tolist()],
                   "chain_failures": [int(x) for x in mine[:, 2].tolist()],
                   "dropped_steps": [int(x) for x in mine[:, 3].tolist()],
                   "fell_back": [bool(x) for x in mine[:, 4].tolist()],
                   "gradient_dtype": "bf16" if getattr(dist, "grad_dtype", None) is not None else "f32",
                   # how many ranks the RCCL communicator itself reports (ncclCommCount through lxo_comm_info): proves N ranks met on RCCL
                   "rccl_ranks_seen": dist.lxo.ranks_seen if getattr(dist, "lxo", None) is not None else None,
                   "data_plane": "RCCL via the C ABI (lxo_allreduce_bucket)" if getattr(dist, "lxo", None) is not None else "torch.distributed (%s)" % td.get_backend()}
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce_max(tt)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)

    if rank == 0:
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.sim else args.dtype, "data": "synthetic",
            "config": {"workload": "configs[2]: full encoder+attention+decoder training step, batch %d/GPU, %dx%d, vocab %d, "
                                   "T=%d (lengths U{30..100}), Adam" % (B, H, W, V, T),
                       "global_batch": B * world, "parallelism": "dp%d" % world},
        }
        if dp_info is not None:
            out["data_parallel"] = dict(dp_info, rank_cpu_pinning=(pinned if not args.sim else None))
        if args.sim:
            out["config"]["workload"] = "TEST ONLY (--sim): hipsim-interpreted kernels on CPU, gloo, tiny shapes"
        if not args.sim and not args.no_extras:
            out["conv_roofline_fraction_e2e"] = round(GF_TRAIN_PER_IMG * (H * W / (128.0 * 512.0)) * value / world / MFMA_BF16_PEAK, 4)
            recs, phases = instrumented_step(eng, img, f_d, l, torch)
            roof = roofline_from_records(recs, ("conv_fwd", "conv_dgrad"),
                                         "conv_halo2wg_kernel (bf16 implicit-GEMM 3x3 conv, halo tiles): the 10 conv forward / data-gradient launches of a step (conv2/4/5 forward include their fused max pool; FLOPs count the convolution only)",
                                         "mfma", MFMA_BF16_PEAK, "TFLOP/s")
            # HBM traffic measured in this run (two short PMC passes of this same script); the committed file only if rocprofv3 is unusable
            traffic = None
            if world == 1 and not args.no_pmc:
                traffic = pmc_traffic_inrun(["conv_halo2wg_kernel", "conv_wgrad_kernel", "xdec_fwd_kernel", "xdec_bwd_kernel", "attn_bwd_part_kernel", "attn_fwd_part_kernel", "rstep_kernel"])
            if roof is not None:
                if traffic and "conv_halo2wg_kernel" in traffic:
                    roof["traffic"] = traffic["conv_halo2wg_kernel"]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = "this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over 3 training steps of this script, per launch; FETCH x2 (gfx950)"
                else:
                    tpath = os.path.join(ROOT, "profiles", "r04_traffic.json")
                    if os.path.exists(tpath):
                        try:
                            tj = json.load(open(tpath))
                            roof["traffic"] = tj["conv_halo2wg_kernel"]["hbm_bytes_per_launch"]
                            roof["traffic_source"] = "profiles/r04_traffic.json (rocprofv3 was not usable inside this run): the same two PMC passes, taken earlier in round 4"
                        except Exception:
                            pass
            if traffic:
                out["pmc_traffic_per_launch"] = traffic
            out["roofline"] = roof
            out["roofline_wgrad"] = roofline_from_records(recs, ("conv_wgrad",), "conv_wgrad_kernel (bf16 3x3 weight gradient, tap reuse): 5 launches of a step",
                                                          "mfma", MFMA_BF16_PEAK, "TFLOP/s")
            if out["roofline_wgrad"] is not None and traffic and "conv_wgrad_kernel" in traffic:
                out["roofline_wgrad"]["traffic"] = traffic["conv_wgrad_kernel"]["hbm_bytes_per_launch"]
                out["roofline_wgrad"]["traffic_source"] = "this run: the same two rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE per launch)"
            out["roofline_attention"] = roofline_from_records(recs, ("attn_fwd",), "attn_fwd_part_kernel + attn_fwd_combine_kernel: one decoder step, B samples, att_img + img streamed once",
                                                              "hbm", HBM_PEAK, "GB/s")
            if out["roofline_attention"] is None:
                # the forward recurrence is ONE persistent launch (csrc/xdec.hip): the attention stream is its phase P3, timed by in-kernel stamps
                ph = chain_phases(eng, torch)
                chain = [r for r in recs if r[0] == "xdec_fwd"]
                if ph and chain:
                    bytes_step = chain[0][2] / T                     # algorithmic bytes of one step: B * R * (E + C) * 2
                    # the stream's time = phase P3 + the wait behind it: the first blocks of the NEXT step are requested at the end of P3 and
                    # land before that barrier (loads return in order), so a quarter of a step's bytes arrives there
                    stream_us = ph["P3_attention_stream"] + ph["barrier3"] + ph.get("partials_poll_wait", 0.0)      # phase + the hand-over wait behind it
                    a = bytes_step / (stream_us * 1e-6)
                    out["roofline_attention"] = {
                        "kernel": "xdec_fwd_kernel, phase P3 + the wait behind it (attention stream of one decoder step: B samples, att_exp + img streamed once; the other phases of the step are the LSTM / att_h / o-projection GEMMs and four XCD barriers)",
                        "bound": "hbm", "achieved": round(a / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a / HBM_PEAK, 4),
                        "traffic": (traffic["xdec_fwd_kernel"]["hbm_bytes_per_launch"] / T) if traffic and "xdec_fwd_kernel" in traffic else None,
                        "launches": 1, "work_per_launch": bytes_step, "avg_launch_us": round(stream_us, 3),
                        "chain_us_per_step_by_phase": ph, "chain_ms_per_launch": round(chain[0][3] * 1e3, 4),
                        "whole_chain_GBps": round(chain[0][2] / chain[0][3] / 1e9, 2),
                        "source": "in-kernel 100 MHz timestamps (lxo_xdec_debug), mean over 256 workgroups x the steps of one launch; whole-launch time from HIP events (lxo_timing_*); traffic per STEP from the PMC passes of this run"}
            out["roofline_attention_bwd"] = roofline_from_records(recs, ("attn_bwd",), "attn_bwd_part_kernel: one BPTT step, the same two streams",
                                                                  "hbm", HBM_PEAK, "GB/s")
            if traffic and out.get("roofline_attention_bwd") and "attn_bwd_part_kernel" in traffic:
                out["roofline_attention_bwd"]["traffic"] = traffic["attn_bwd_part_kernel"]["hbm_bytes_per_launch"]
            if out["roofline_attention_bwd"] is None:
                # BPTT is ONE persistent launch as well (xdec_bwd_kernel): its attention stream is phase Q2
                ph = chain_phases(eng, torch, backward=True)
                chain = [r for r in recs if r[0] == "xdec_bwd"]
                if ph and chain:
                    bytes_step = chain[0][2] / T
                    stream_us = ph["Q2_attention_stream"] + ph["barrier2"] + ph.get("partials_poll_wait", 0.0)       # as for the forward chain: phase + the wait behind it
                    a = bytes_step / (stream_us * 1e-6)
                    out["roofline_attention_bwd"] = {
                        "kernel": "xdec_bwd_kernel, phase Q2 + the wait behind it (attention stream of one BPTT step: d_e and d_att_h from att_exp + img streamed once; the other phases are the [d_h~ | d_ctx] / d_att_h W^T / carry GEMMs with the LSTM backward and four XCD barriers)",
                        "bound": "hbm", "achieved": round(a / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(a / HBM_PEAK, 4),
                        "traffic": (traffic["xdec_bwd_kernel"]["hbm_bytes_per_launch"] / T) if traffic and "xdec_bwd_kernel" in traffic else None,
                        "launches": 1, "work_per_launch": bytes_step, "avg_launch_us": round(stream_us, 3),
                        "chain_us_per_step_by_phase": ph, "chain_ms_per_launch": round(chain[0][3] * 1e3, 4),
                        "whole_chain_GBps": round(chain[0][2] / chain[0][3] / 1e9, 2),
                        "source": "in-kernel 100 MHz timestamps (lxo_xdec_debug_bwd), as for the forward chain"}
            for key in ("roofline_attention", "roofline_attention_bwd"):
                r = out.get(key)
                if r and r.get("whole_chain_GBps"):
                    # `frac` above prices the STREAM PHASE of a step (P3 / Q2 + the wait behind it); the launch as a whole also runs the step's serial
                    # phases (LSTM / att_h / o-projection GEMMs, hand-overs), during which nothing streams: algorithmic bytes of all T steps / launch time
                    r["whole_launch_frac"] = round(r["whole_chain_GBps"] * 1e9 / HBM_PEAK, 4)
                    r["frac_is"] = "stream phase only; whole_launch_frac = the same bytes over the whole persistent launch"
                if r and r.get("traffic") and r.get("avg_launch_us"):
                    # `achieved` counts ALGORITHMIC bytes (every region row of att_exp + img once per step); consecutive steps walk the chunk in
                    # opposite directions, so part of them is still in the XCD's L2 -- the PMC traffic is what HBM / Infinity Cache delivered
                    r["hbm_GBps_from_traffic"] = round(r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9, 1)
                    r["note"] = ("achieved = algorithmic bytes / stream time; %.0f %% of them are L2 hits (traffic < work_per_launch): "
                                 "the memory system delivered hbm_GBps_from_traffic" % (100.0 * (1.0 - r["traffic"] / r["work_per_launch"])))
            out["ms_per_step_by_phase"] = phases
            out["ms_per_step_by_phase_note"] = ("one instrumented step with every launch on one stream; the timed steps overlap the weight-gradient "
                                                "launches (encoder, and the decoder's deferred ones) with the data-gradient path on a second stream, "
                                                "so the phases sum to more than ms_per_step")
            if world == 1:
                if not args.no_secondary:
                    try:
                        out["secondary"] = secondary(args, eng, torch, dev, B, H, W, V)
                    except Exception as e:          # never lose the headline line to a secondary measurement
                        out["secondary"] = {"error": repr(e)}
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
                if not args.no_secondary:
                    try:
                        out.update(cpu_baselines_secondary(eng, torch, dev, V))
                    except Exception as e:
                        out["cpu_baselines_secondary_error"] = repr(e)
        # librccl prints a version banner through C stdio at its first communicator (secondary.data_parallel_world1, or the N > 1 run itself): on a pipe it
        # would sit in the C buffer until exit and land BEHIND the JSON line -- flush it out first, so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        import torch.distributed as td
        dist.barrier()                   # ranks > 0 wait for rank 0's extra measurements before tearing down
        dist.close()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
