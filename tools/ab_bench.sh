#!/bin/bash
# usage (GPU box): bash tools/ab_bench.sh OUT "<args A>" "<args B>" [rounds]: bench.py with two argument sets, alternating on the
# SAME box (boxes differ by a few per cent, so only same-box pairs compare) -> gpurun_out/OUT_ab.txt
cd "$GRAFT_REPO_ROOT"
rounds=${4:-3}
: > gpurun_out/$1_ab.txt
for i in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then args="$2"; else args="$3"; fi
    python bench.py --steps 150 --warmup 6 --meter-frames 0 --h2d-steps 0 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$args', round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/$1_ab.txt
  done
done
cat gpurun_out/$1_ab.txt
