"""Family summary + per-launch list of a tools/seq_from_trace.py sequence, as committed under profiles/.
usage: python tools/seq_summary.py gpurun_out/X_seq.txt "header line" > profiles/rNN_decoder_graph_alone.txt"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
tot, cnt = collections.Counter(), collections.Counter()
rows = []
for l in lines:
    m = re.match(r"\s*(-?[\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(.*)", l)
    if not m:
        continue
    name = m.group(4).split("  grid=")[0]
    fam = re.sub(r"<.*", "", name)
    fam = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", fam)
    fam = re.sub(r"(IDF16_|Ef|EP|EvP).*", "", fam)
    tot[fam] += float(m.group(2))
    cnt[fam] += 1
    rows.append(l)
print(sys.argv[2])
print(f"{sum(cnt.values())} kernels, {sum(tot.values()):.0f} us busy")
for fam, t in tot.most_common():
    print(f"{t:8.1f} us {cnt[fam]:4d}x {fam}")
print("\nper launch, in order (start offset us, duration us, gap us, kernel):")
print("\n".join(rows))
