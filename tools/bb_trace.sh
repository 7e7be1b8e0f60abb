#!/bin/bash
# usage (GPU box, repo root): bash tools/bb_trace.sh OUT  -> gpurun_out/OUT_bbseq.txt: per-launch sequence of one replayed backbone graph
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
d=gpurun_out/_prof_$1
rocprofv3 --kernel-trace --output-format csv -d $d -o bb -- python3 tools/stream_times.py --bb-only ${@:2} > gpurun_out/$1_bbtrace.log 2>&1
f=$(find $d -name "*kernel_trace.csv" | tail -1)
python tools/seq_from_trace.py $f stem_conv_pool > gpurun_out/$1_bbseq.txt
rm -rf $d
tail -1 gpurun_out/$1_bbseq.txt
