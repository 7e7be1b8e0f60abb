"""Per-launch kernel sequence of the LAST decoder-graph replay in a rocprofv3 --kernel-trace csv of
`tools/stream_times.py --dec-only` (or any run whose decoder queue starts a frame with bank/alloc kernels).
usage: seq_from_trace.py trace.csv [marker-substring]   -> one line per launch: start offset, duration, gap, name"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "decode3d"
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = marks[-2] + 1, marks[-1] + 1
q = rows[marks[-1]]["Queue_Id"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n.split("(")[0][:70]


t0 = None
prev_end = None
tot = 0.0
for r in rows[lo:hi]:
    if r["Queue_Id"] != q:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} {gap:5.1f}  {short(r['Kernel_Name'])}  grid={r.get('Grid_Size_X', '?')} wg={r.get('Workgroup_Size_X', '?')}")
    prev_end = e
    tot += (e - s) / 1e3
print(f"busy {tot:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
