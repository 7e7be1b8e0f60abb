"""Per-frame kernel breakdown from a rocprofv3 --kernel-trace csv of bench.py: takes the last
frames (graph replays), groups kernels by name. usage: frame_breakdown.py trace.csv [frames]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "alloc_project" in r["Kernel_Name"]]
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# a frame = from one 'first alloc_project of a frame' to the next; 3 alloc_project per frame
firsts = marks[0::3]
lo, hi = firsts[-nf - 1], firsts[-1]
sub = rows[lo:hi]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sub:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[r["Kernel_Name"][:86]][0] += 1
    agg[r["Kernel_Name"][:86]][1] += d
busy = sum(v[1] for v in agg.values())
span = (int(sub[-1]["End_Timestamp"]) - int(sub[0]["Start_Timestamp"])) / 1e3
print(f"{nf} frames: {len(sub) / nf:.0f} kernels/frame, busy {busy / nf / 1e3:.2f} ms/frame, span {span / nf / 1e3:.2f} ms/frame")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k:86s} n/frame={v[0] / nf:6.1f} us/frame={v[1] / nf:8.1f} avg_us={v[1] / v[0]:7.1f}")
