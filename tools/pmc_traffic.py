#!/usr/bin/env python
"""Turns the two PMC passes (FETCH_SIZE, WRITE_SIZE; rocprofv3 rocpd sqlite) over a few real training steps
(`bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-secondary`, see tools/gpu_call.sh `traffic`) into
profiles/r02_conv_traffic.json.  Units/corrections per MI355X_MICROARCH.md section HBM: both counters are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is doubled; WRITE_SIZE is taken
as reported (uncalibrated).  Averages are per launch over every dispatch of the conv kernel in the workload
(the 10 launch shapes of a step x the steps run)."""
import json, sqlite3, sys


def avg(dbpath, counter, pat):
    db = sqlite3.connect(dbpath)
    q = ("select count(*), avg(value) from counters_collection where kernel_name like ? and counter_name = ?")
    n, v = db.execute(q, ("%" + pat + "%", counter)).fetchone()
    return n, v


def main():
    fdb, wdb, out = sys.argv[1], sys.argv[2], sys.argv[3]
    pat = sys.argv[4] if len(sys.argv) > 4 else "conv_halo2wg_kernel"
    nf, f = avg(fdb, "FETCH_SIZE", pat)
    nw, w = avg(wdb, "WRITE_SIZE", pat)
    res = {"kernel": pat, "dispatches": nf, "fetch_size_kb_avg_raw": f, "write_size_kb_avg_raw": w,
           "fetch_bytes_per_launch": 2.0 * f * 1024.0, "write_bytes_per_launch": w * 1024.0,
           "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
           "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-secondary"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
