#!/usr/bin/env python
"""Short view of a bench.py JSON line: headline, rooflines, per-launch conv rates, phase table."""
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print("value %.1f %s, %.3f ms/step, n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
for k in ("roofline", "roofline_wgrad", "roofline_attention", "roofline_attention_bwd"):
    r = d.get(k)
    if r:
        print("%-24s %8.1f %-8s frac %.4f  avg %.2f us x %d" % (k, r["achieved"], r["unit"], r["frac"], r["avg_launch_us"], r["launches"]))
        if k in ("roofline", "roofline_wgrad"):
            print("    " + "  ".join("%s %.0fus/%.0f" % (kk.split(":")[0][5:] + ":" + kk.split(":")[1], v["us"], v["rate"]) for kk, v in r["per_launch"].items()))
print(d.get("ms_per_step_by_phase"))
