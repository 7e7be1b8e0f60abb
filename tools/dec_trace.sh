#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/dec_trace.sh OUT_PREFIX  -> gpurun_out/OUT_PREFIX_seq.txt (per-launch
# sequence of the replayed decoder graph, nothing beside it) from a rocprofv3 --kernel-trace of tools/stream_times.py --dec-only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/_prof_$1
rocprofv3 --kernel-trace --output-format csv -d $d -o dec -- python3 tools/stream_times.py --dec-only ${@:2} > gpurun_out/$1_trace.log 2>&1
f=$(find $d -name "*kernel_trace.csv" | tail -1)
python tools/seq_from_trace.py $f > gpurun_out/$1_seq.txt
rm -rf $d
tail -1 gpurun_out/$1_seq.txt
