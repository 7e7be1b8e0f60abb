"""Summaries of the rocprofv3 passes behind the roofline numbers, written under profiles/ (run AFTER the passes):

    cd /tmp && export TMPDIR=/tmp      # on the GPU box, from the repo root:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stats -- python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof/fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof/write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline
    python3 tools/profile_round.py gpurun_out/prof profiles/r02      # -> profiles/r02_bench_kernel_stats.csv, r02_sampler_traffic.json

(separate --pmc passes, no tracing domains beside --kernel-trace: the counter guide's recipe). The same three passes around
`python3 tools/bench_daf_hbm.py` (R101 1408x512 feature set, 368 MB, rotated buffers: every launch cold) with the out prefix
profiles/r02_daf_r101_cold give the out-of-cache figure.
"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {"daf_fwd_rows": "daf_fwd_rows", "msda_grouped_fwd": "msda_grouped_fwd", "conv1x1_f16_kernel": "conv1x1_f16",
           "linear_f16x3_kernel": "linear_f16x3", "format_tokens_kernel": "format_tokens", "mlp_chain_mfma_kernel": "mlp_chain_mfma",
           "gemm_f16x3_wide_kernel": "gemm_f16x3_wide", "gemm_f16x3_kernel": "gemm_f16x3", "attention_halfs_kernel": "attention_halfs",
           "gemm_f32_kernel": "gemm_f32", "attention_f32_kernel": "attention_f32", "mlp_chain_r4_kernel": "mlp_chain_r4",
           "conv_staged_kernel": "conv_staged (3x3 / 1x1)", "conv3x3_f16_kernel": "conv3x3 direct", "linear_h2_kernel": "linear_h2 (value_proj)",
           "alloc_static_kernel": "alloc_static", "daf_fused_rows": "daf_fused_rows", "msda_linear_fwd": "msda_linear_fwd",
           "stem_conv_pool_kernel": "stem_conv_pool"}


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    return hits[-1] if hits else None


def kernel_stats(stats_dir):
    """rows of <pid>_kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev)."""
    path = find(stats_dir, "*kernel_stats.csv")
    if path is None:
        return None, []
    with open(path) as f:
        return path, list(csv.DictReader(f))


def counter_avg(pass_dir, counter):
    """{short kernel name: (mean counter value per dispatch, dispatches)} from <pid>_counter_collection.csv."""
    path = find(pass_dir, "*counter_collection.csv")
    agg = collections.defaultdict(list)
    if path is None:
        return agg
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            for k, short in KERNELS.items():
                if k in r["Kernel_Name"]:
                    agg[short].append(float(r["Counter_Value"]))
                    break
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    src, out = sys.argv[1], sys.argv[2]
    command = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps N --warmup W --no-cpu-baseline"
    path, rows = kernel_stats(os.path.join(src, "stats"))
    stats = {}
    if rows:
        keep = [r for r in rows if float(r.get("Percentage", 0) or 0) >= 0.05 or any(k in r["Name"] for k in KERNELS)]
        with open(out + "_bench_kernel_stats.csv" if "daf_r101" not in out else out + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
        for r in rows:
            for k, short in KERNELS.items():
                if k in r["Name"] and short not in stats:
                    stats[short] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
                                        max_us=float(r["MaxNs"]) / 1e3)
    fetch, write = counter_avg(os.path.join(src, "fetch"), "FETCH_SIZE"), counter_avg(os.path.join(src, "write"), "WRITE_SIZE")
    kernels = {}
    for short in sorted(set(fetch) | set(write) | set(stats)):
        f_kb, fn = fetch.get(short, (None, 0))
        w_kb, wn = write.get(short, (None, 0))
        entry = dict(rocprof=stats.get(short), pmc=dict(FETCH_SIZE_KB_avg=f_kb, FETCH_SIZE_n=fn, WRITE_SIZE_KB_avg=w_kb, WRITE_SIZE_n=wn))
        if f_kb is not None and w_kb is not None:
            entry["traffic_bytes_per_launch"] = (2 * f_kb + w_kb) * 1024
            if stats.get(short):
                entry["hbm_side_GBps_at_rocprof_avg"] = entry["traffic_bytes_per_launch"] / (stats[short]["avg_us"] * 1e-6) / 1e9
        kernels[short] = entry
    json.dump(dict(
        command=f"rocprofv3 --kernel-trace --stats / --kernel-trace --pmc FETCH_SIZE / --kernel-trace --pmc WRITE_SIZE (three separate passes) -- {command}",
        round=int(os.path.basename(out)[1:3]) if os.path.basename(out)[1:3].isdigit() else None,
        correction="traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KB; gfx950 FETCH_SIZE counts half of the "
                   "bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included in FETCH_SIZE; "
                   "averages over all dispatches of a kernel name in the pass",
        kernels=kernels), open(out + "_sampler_traffic.json" if "daf_r101" not in out else out + ".json", "w"), indent=1)
    for k, v in kernels.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
