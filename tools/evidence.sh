#!/bin/bash
# usage (GPU box, repo root): bash tools/evidence.sh rNN [quick]  -> gpurun_out/rNN_*: the round's bench lines, rocprofv3 kernel
# statistics, PMC traffic and MFMA utilisation of the final tree. Copy what is to be judged into profiles/ afterwards
# (everything lands under gpurun_out/, the only directory that travels back).
R=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
if [ "$2" != prof ]; then
python bench.py > $O/${R}_bench_n1_default_flags.json 2> $O/${R}_bench_default.err && echo "default: $(python -c "import json;d=json.load(open('$O/${R}_bench_n1_default_flags.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_msda']['frac'], d['cpu_baseline']['value'])")"
python bench.py --steps 100 --warmup 6 --no-cpu-baseline > $O/${R}_bench_n1.json 2> /dev/null && echo "n1: $(python -c "import json;d=json.load(open('$O/${R}_bench_n1.json'));print(d['value'], d['ms_per_step'])")"
python bench.py --steps 100 --warmup 6 --no-cpu-baseline --streams 8 > $O/${R}_bench_n1_8streams.json 2> /dev/null && echo "8 streams: $(python -c "import json;d=json.load(open('$O/${R}_bench_n1_8streams.json'));print(d['value'], d['ms_per_step'])")"
python bench.py --steps 40 --warmup 6 --no-cpu-baseline --bs 8 > $O/${R}_bench_n1_bs8_independent_streams.json 2> /dev/null && echo "bs 8 (independent streams): $(python -c "import json;d=json.load(open('$O/${R}_bench_n1_bs8_independent_streams.json'));print(d['value'], d['ms_per_step'])")"
python bench.py --steps 40 --warmup 6 --no-cpu-baseline --bs 8 --reference-batch > $O/${R}_bench_n1_bs8_reference_batch.json 2> /dev/null && echo "bs 8 (reference batch): $(python -c "import json;d=json.load(open('$O/${R}_bench_n1_bs8_reference_batch.json'));print(d['value'], d['ms_per_step'])")"
python bench.py --steps 100 --warmup 6 --no-cpu-baseline --h2d > $O/${R}_bench_n1_h2d.json 2> /dev/null && echo "h2d: $(python -c "import json;d=json.load(open('$O/${R}_bench_n1_h2d.json'));print(d['value'], d['ms_per_step'])")"
python bench.py --steps 60 --warmup 6 --no-cpu-baseline --depth 101 --image-wh 1408 512 --residual-damp 0.3 --token-std 798.5 > $O/${R}_bench_r101_1408x512.json 2> $O/${R}_bench_r101.err && echo "r101: $(python -c "import json;d=json.load(open('$O/${R}_bench_r101_1408x512.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])")"
python tools/stream_times.py 2>&1 | tail -3 > $O/${R}_stream_times.txt; cat $O/${R}_stream_times.txt
python tools/stream_times.py --bs 8 2>&1 | tail -3 > $O/${R}_stream_times_bs8.txt; cat $O/${R}_stream_times_bs8.txt
fi
[ "$2" = quick ] && exit 0
P=$O/prof_$R
rm -rf $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/${R}_bench_under_rocprof_line.json 2> $O/${R}_prof_stats.log
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/${R}_prof_fetch.log
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/${R}_prof_write.log
python3 tools/profile_round.py $P $O/$R > $O/${R}_profile_round.log 2>&1; tail -4 $O/${R}_profile_round.log
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $P/mfma -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/${R}_prof_mfma.log
f=$(find $P/mfma -name "*counter_collection.csv" | tail -1); python3 tools/mfma_util.py $f $O/${R}_mfma_util.json > $O/${R}_mfma_util.log 2>&1; tail -3 $O/${R}_mfma_util.log
rm -rf $P
bash tools/dec_trace.sh ${R}_dec; bash tools/bb_trace.sh ${R}; bash tools/dec_trace.sh ${R}_dec_bs8 --bs 8
python3 tools/bench_daf_hbm.py > $O/${R}_daf_r101_cold_tool_line.json 2>/dev/null; cat $O/${R}_daf_r101_cold_tool_line.json
SIMPB_BENCH_DEVICE=0 python bench.py --gpus 2 --backend gloo --capacity-by-rank 1536 256 --steps 20 --warmup 4 --no-cpu-baseline > $O/${R}_bench_n2_rehearsal_gloo_one_gpu.json 2> $O/${R}_bench_n2.err; tail -c 400 $O/${R}_bench_n2_rehearsal_gloo_one_gpu.json
# victim-side check of the per-file NO_PACKED_FP32 build (simpb_amd/build.py): the library as shipped, only the co-runner's matrix step on the double-K instruction
: > $O/${R}_victims_shipped_flags.log
for v in daf layernorm attention chain gemm; do timeout -k 10 200 python tools/daf_stress.py --k16 --co conv1x1 --launches 1000 --victim $v 2>&1 | tail -1 >> $O/${R}_victims_shipped_flags.log; done
cat $O/${R}_victims_shipped_flags.log
