"""Per-kernel averages of a rocprofv3 --pmc counter_collection csv: usage: pmc_table.py file.csv [name-substring]"""
import collections
import csv
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ""
per = collections.defaultdict(dict)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            d = per[(r["Kernel_Name"][:90], r["Dispatch_Id"])]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, _), c in per.items():
    for n, v in c.items():
        agg[k][n].append(v)
for k, c in agg.items():
    print(k)
    m = {n: sum(v) / len(v) for n, v in c.items()}
    for n, v in sorted(m.items()):
        print(f"    {n:32s} {v:16.1f}")
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print(f"    mfma_busy_fraction               {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
    if "SQ_WAVE_CYCLES" in m:
        for n in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if n in m:
                print(f"    {n}/WAVE_CYCLES      {m[n] / m['SQ_WAVE_CYCLES']:.3f}")
