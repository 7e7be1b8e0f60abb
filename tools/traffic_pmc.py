"""HBM-side traffic of selected kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
the microarchitecture guide prescribes). Streams the csv files (hundreds of MB).
usage: traffic_pmc.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import sys

NAMES = {"conv1x1_f16_kernel": "conv1x1_f16", "linear_f16x3_kernel": "linear_f16x3 (value_proj)", "daf_fwd_rows": "daf_fwd_rows",
         "msda_grouped_fwd": "msda_grouped_fwd", "format_tokens_kernel": "format_tokens"}


def collect(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            for k, v in NAMES.items():
                if k in r["Kernel_Name"]:
                    agg[v].append(float(r["Counter_Value"]))
                    break
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kb = sum(fetch[k]) / max(len(fetch[k]), 1)
    w_kb = sum(write[k]) / max(len(write[k]), 1)
    out[k] = dict(dispatches=len(fetch[k]), FETCH_SIZE_KB_avg=round(f_kb, 1), WRITE_SIZE_KB_avg=round(w_kb, 1),
                  traffic_MB_per_launch=round((2 * f_kb + w_kb) * 1024 / 1e6, 2))
    print(k, out[k])
json.dump(dict(command="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 6 "
                       "--warmup 2 --no-cpu-baseline", round=1,
               correction="traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count for 16 B/lane reads; Infinity-Cache "
                          "hits are included in FETCH_SIZE); averages over all dispatches of a kernel name (all shapes)",
               kernels=out), open(sys.argv[3], "w"), indent=1)
