import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import *
from tests.test_gpu_head import run_product_stream
g = load_golden(sys.argv[1] if len(sys.argv) > 1 else "head_r50.npz")
spec = spec_of(g)
for f, trace, outs, res, head in run_product_stream(g):
    pre = f"f{f}."
    print("frame", f, "n2", [x.shape[1] for x in outs["prediction2d"]], g[pre + "n2#0"].tolist())
    a = head.instance_bank.cached_anchor.cpu().numpy(); b = g[pre + "bank.cached_anchor#0"]
    d = np.abs(a - b).max(-1)[0]
    print("  cached_anchor rows differing >1e-3:", np.where(d > 1e-3)[0].tolist()[:20], "max", d.max())
    ca = head.instance_bank.confidence.cpu().numpy()[0]; cb = g[pre + "bank.confidence#0"][0]
    bad = np.where(d > 1e-3)[0]
    for r in bad[:6]:
        print("   row", r, "conf got", ca[r], "want", cb[r], "neighbors want", cb[max(r-1,0):r+2])
    if f in spec["trace_frames"]:
        try:
            compare_trace(trace, g, pre + "trace.", rtol=1e-3, atol=1e-3)
            print("  trace ok")
        except AssertionError as e:
            print("  trace:", str(e)[:600])
    for b_, r in enumerate(res):
        try:
            compare_result(r["img_bbox"], g, f"{pre}res{b_}.")
            print("  result ok")
        except AssertionError as e:
            print("  result:", str(e)[:300])
