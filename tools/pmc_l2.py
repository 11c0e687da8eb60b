#!/usr/bin/env python
"""L2 hit rate per kernel from one `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace` pass over a few training steps
(`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-secondary`): sums over every dispatch of each named kernel.
usage: pmc_l2.py <results.db> <out.json> [kernel name patterns ...]"""
import json, sqlite3, sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pats = sys.argv[3:] or ["xdec_fwd_kernel", "attn_bwd_part_kernel", "rstep_kernel", "conv_halo2wg_kernel", "conv_wgrad_kernel"]
    out = {}
    for pat in pats:
        row = {}
        for c in ("TCC_HIT_sum", "TCC_MISS_sum"):
            n, v = db.execute("select count(*), sum(value) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + pat + "%", c)).fetchone()
            row[c] = v; row["dispatches"] = n
        if row.get("TCC_HIT_sum") is not None and row.get("TCC_MISS_sum") is not None and (row["TCC_HIT_sum"] + row["TCC_MISS_sum"]) > 0:
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
        out[pat] = row
    out["note"] = "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-secondary; requests of 128 bytes, summed over the 16 channels of the 8 XCDs and over all dispatches"
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
