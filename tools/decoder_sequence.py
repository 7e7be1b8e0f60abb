"""Kernel sequence of one replayed frame on the decoder's stream, from a rocprofv3 --kernel-trace csv
of bench.py. usage: decoder_sequence.py trace.csv [--all]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "alloc_project" in r["Kernel_Name"]]
q = rows[marks[-4]]["Queue_Id"]
lo, hi = marks[-4], marks[-1]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|launch_clamp\w*|\w+Ops)", n)
    base = n.split("<")[0].split("(")[0]
    if base in ("vectorized_elementwise_kernel", "elementwise_kernel_manual_unroll", "elementwise_kernel",
                "reduce_kernel", "unrolled_elementwise_kernel") and m:
        return base[:12] + ":" + m.group(1)
    return base[:50]


seq = []
for r in rows[lo:hi]:
    if r["Queue_Id"] != q:
        continue
    seq.append((short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
ours = ("gemm_f32", "attention_f32", "mlp_chain", "layernorm_seg", "daf_fwd", "msda_", "dfa_", "alloc_", "gather_rows",
        "aggregate_kernel", "anchor_projection", "rowdot", "fill_int", "linear_f32", "format_tokens")
print(f"{len(seq)} kernels, {sum(d for _, d in seq):.0f} us busy on the decoder stream")
mine = sum(d for n, d in seq if any(n.startswith(o) for o in ours))
print(f"  own kernels {mine:.0f} us in {sum(1 for n, _ in seq if any(n.startswith(o) for o in ours))} launches; "
      f"others {sum(d for _, d in seq) - mine:.0f} us")
if "--all" in sys.argv:
    for n, d in seq:
        print(f"{d:6.1f} {n}")
else:
    import collections
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, d in seq:
        agg[n][0] += 1
        agg[n][1] += d
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{d:8.1f} us {c:4d}x {n}")
