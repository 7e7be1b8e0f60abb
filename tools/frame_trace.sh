#!/bin/bash
# usage (GPU box, repo root): bash tools/frame_trace.sh OUT [bench args] -> gpurun_out/OUT_frame.txt: every kernel of the last
# replayed frames of `bench.py` (all queues, in start order, with queue id) from a rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/_prof_$1
rocprofv3 --kernel-trace --output-format csv -d $d -o fr -- python3 bench.py --steps 12 --warmup 4 --meter-frames 0 --no-cpu-baseline ${@:2} > gpurun_out/$1_frame.log 2>&1
f=$(find $d -name "*kernel_trace.csv" | tail -1)
python3 - "$f" > gpurun_out/$1_frame.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "decode3d" in r["Kernel_Name"]]
lo, hi = marks[-3] + 1, marks[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[lo:hi]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:48]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {'    ' * q}{n}")
PY
rm -rf $d
wc -l gpurun_out/$1_frame.txt
