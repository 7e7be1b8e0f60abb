"""Timeline of one replayed frame on the decoder's stream from a rocprofv3 --kernel-trace csv of bench.py: start offset,
duration and the idle gap in front of every kernel, then totals per kernel family (busy time, gaps charged to the kernel
that follows). usage: frame_timeline.py trace.csv [--all]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "alloc_project" in r["Kernel_Name"]]
q = rows[marks[-4]]["Queue_Id"]
lo, hi = marks[-7], marks[-4]      # alloc_project runs three times per frame: one whole frame between the marks
if "--frames" in sys.argv:         # average over the last N frames instead
    nf = int(sys.argv[sys.argv.index("--frames") + 1])
    lo, hi = marks[-3 * nf - 1], marks[-1]
else:
    nf = 1


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n.split("(")[0][:60]


seq = [r for r in rows[lo:hi] if r["Queue_Id"] == q]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
busy = gaps = 0.0
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end) / 1e3
    d = (e - s) / 1e3
    name = short(r["Kernel_Name"])
    if "--all" in sys.argv:
        print(f"{(s - t0) / 1e3:8.1f} us  +{gap:5.1f} gap  {d:6.1f} us  {name}  grid {r.get('Grid_Size', '')} wg {r.get('Workgroup_Size', '')}")
    busy += d
    gaps += gap
    f = fam[name.split("<")[0]]
    f[0] += 1
    f[1] += d
    f[2] += gap
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
print(f"{len(seq) / nf:.0f} kernels per frame on the decoder queue: span {span / nf:.0f} us = busy {busy / nf:.0f} us + gaps {gaps / nf:.0f} us")
for n, (c, d, g) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{d / nf:8.1f} us busy {g / nf:7.1f} us gaps in front {c / nf:6.1f}x {n}")
