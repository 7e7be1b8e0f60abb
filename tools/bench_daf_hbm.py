"""DAF / MSDA samplers when the feature set does NOT fit the 256 MiB Infinity Cache: R101 1408x512
shapes (359 040 tokens = 368 MB fp32), rotating over several feature buffers so every launch starts
cold. Prints algorithmic GB/s (SURVEY.md 8d bytes) per kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simpb_amd import synth
from simpb_amd.plugin import ops
from tools.bench_ops import realistic_daf_inputs

wh = (1408, 512)
shapes = synth.level_shapes(wh)
tokens = 6 * sum(h * w for h, w in shapes)
nbuf = 3
feats = [torch.randn(1, tokens, 256, device="cuda") for _ in range(nbuf)]
ss = torch.tensor([shapes] * 6, dtype=torch.int32, device="cuda")
sizes = [h * w for h, w in shapes] * 6
ssi = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]).reshape(6, 4), dtype=torch.int32, device="cuda")
loc, w = realistic_daf_inputs(1, wh)
valid = int(((loc > 0) & (loc < 1)).all(-1).sum())
nbytes = valid * 4 * 4 * 256 * 4 + loc.numel() * 4 + w.numel() * 4 + 900 * 256 * 4
for _ in range(3):
    for f in feats:
        ops.deformable_aggregation_function(f, ss, ssi, loc, w)
torch.cuda.synchronize()
evs = []
for it in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.deformable_aggregation_function(feats[it % nbuf], ss, ssi, loc, w); b.record()
    evs.append((a, b))
torch.cuda.synchronize()
t = np.median([a.elapsed_time(b) for a, b in evs]) * 1e-3
print(json.dumps(dict(kernel="daf_fwd_rows", regime="feature set 368 MB x3 rotating (HBM)", valid_triples=valid,
                      us=t * 1e6, algorithmic_MB=nbytes / 1e6, GBps=nbytes / t / 1e9, frac_of_8TBps=nbytes / t / 8e12)))
