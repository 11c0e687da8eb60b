"""Derived figures from a tools/pmc_dump.py table (SQ counters per kernel instantiation and grid): MFMA pipe busy, issue activity, LDS conflicts.
python tools/sq_summary.py <table.txt> [resident waves, default 2048]"""
import re, sys
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\S.*?)\s+grid\s+(\d+)\s+(\S+)\s+n\s+(\d+)\s+avg\s+(\S+)", line)
    if m:
        rows.setdefault((m.group(1).strip(), int(m.group(2))), {})[m.group(3)] = (int(m.group(4)), float(m.group(5)))
res = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
print("# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (4 x SQ_WAVE_CYCLES / %d resident waves): share of a resident wave's time in which its SIMD's matrix pipe is busy" % res)
for (k, g), c in sorted(rows.items(), key=lambda x: (x[0][1], x[0][0])):
    if "SQ_WAVE_CYCLES" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    wc = c["SQ_WAVE_CYCLES"][1]
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 1024.0 / (4.0 * wc / res)
    act = c.get("SQ_ACTIVE_INST_ANY", (0, 0))[1] / wc
    ldc = c.get("SQ_LDS_BANK_CONFLICT", (0, 0))[1] / wc
    print("#   %-40s grid %8d (%5d workgroups) x%-2d  MFMA pipe busy %5.1f %%   instruction issue active %4.1f %% of wave cycles   LDS bank-conflict cycles / wave cycles %.2f %%" % (
        k, g, g // 256, c["SQ_WAVE_CYCLES"][0], 100 * busy, 100 * act, 100 * ldc))
