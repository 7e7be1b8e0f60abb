"""MFMA utilisation of the matrix kernels from a rocprofv3 --pmc counter_collection csv (streamed: the
file can be hundreds of MB). usage: mfma_util.py counter_collection.csv out.json"""
import collections
import csv
import json
import sys

NAMES = {"gemm_f16x3_wide_kernel": "gemm_f16x3_wide_kernel (64x128 tiles)", "gemm_f16x3_kernel<64": "gemm_f16x3_kernel<64,2,64> (split operands)",
         "gemm_f16x3_kernel<32": "gemm_f16x3_kernel<32,2,128> (split operands)", "attention_halfs_kernel<false": "attention_halfs_kernel<false> (split operands)",
         "attention_halfs_kernel<true": "attention_halfs_kernel<true> (split operands, grouped)",
         "gemm_f32_kernel<64": "gemm_f32_kernel<64,4,64>", "gemm_f32_kernel<32, 2": "gemm_f32_kernel<32,2,128>",
         "gemm_f32_kernel<32, 4": "gemm_f32_kernel<32,4,64>", "attention_f32_kernel<false": "attention_f32_kernel<false>",
         "attention_f32_kernel<true": "attention_f32_kernel<true>", "mlp_chain_mfma": "mlp_chain_mfma_kernel",
         "mlp_chain_r4": "mlp_chain_r4_kernel", "mlp_chain_r32": "mlp_chain_r32_kernel", "linear_f32_mfma": "linear_f32_mfma", "linear_f16x3": "linear_f16x3 (value_proj)",
         "conv1x1_f16": "conv1x1_f16", "conv_staged_kernel<96, 128": "conv_staged<96,128> (FPN 3x3)", "conv_staged_kernel<96, 64": "conv_staged<96,64> (layer1 3x3)",
         "conv_staged_kernel<128, 64": "conv_staged<128,64> (1x1, Cin >= 512)", "conv3x3_f16_kernel": "conv3x3 direct", "linear_h2": "linear_h2 (value_proj)", "stem_conv_pool": "stem_conv_pool (7x7 stem + pool)"}
per = collections.defaultdict(dict)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        for k, v in NAMES.items():
            if k in r["Kernel_Name"]:
                d = per[(v, r["Dispatch_Id"])]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                break
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, _), c in per.items():
    if c.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        agg[k]["util"].append(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))
        agg[k]["dur"].append(c["dur"])
        w = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        agg[k]["stall"].append(c.get("SQ_WAIT_INST_ANY", 0) / w)
        agg[k]["wait"].append(c.get("SQ_WAIT_ANY", 0) / w)
out = {}
for k, v in agg.items():
    n = len(v["util"])
    out[k] = dict(dispatches=n, mfma_busy_fraction=round(sum(v["util"]) / n, 4), avg_us_under_pmc=round(sum(v["dur"]) / n, 2),
                  wave_cycles_in_issue_stall=round(sum(v["stall"]) / n, 3), wave_cycles_waiting=round(sum(v["wait"]) / n, 3))
    print(k, out[k])
json.dump(dict(
    command="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE "
            "-- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline",
    definition="mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) per dispatch, averaged "
               "(the guide's MfmaUtil): the share of SIMD cycles with a matrix instruction in flight, whatever its flop rate "
               "(v_mfma_f32_4x4x1f32 of mlp_chain_r4 keeps the pipe busy at a quarter of the 16x16x4 flop rate). "
               "Counters serialise the two HIP streams of the pipelined frame, so durations are per-kernel, not in-frame.",
    kernels=out), open(sys.argv[2], "w"), indent=1)
