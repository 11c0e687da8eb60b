"""CPU, 2 ranks over gloo: `bench.py --gpus 2` launches its own ranks (torch.distributed.run), runs the data-parallel
training step (device-side token count, bucketed gradient all-reduce hooks) and rank 0 prints ONE JSON line whose
n_gpus equals --gpus.  --sim swaps RCCL for gloo and the gfx950 kernels for their hipsim build (TEST ONLY)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_two_ranks_and_reports_them():
    from simlib import build_sim
    build_sim()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sim", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [x for x in r.stdout.decode().splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["scaling"] == "weak" and out["steps"] == 1 and out["value"] > 0
