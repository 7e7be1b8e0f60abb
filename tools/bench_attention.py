"""Attention core alone: exact-fp32 matrix instruction vs FP16 matrix cores with split operands, the decoder's shapes.
usage: python tools/bench_attention.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import ops  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    rs = np.random.RandomState(0)
    for name, nq, nk, grouped in [("gnn 900 x 900", 900, 900, False), ("temp_gnn 900 x 600", 900, 600, False), ("qg_self_attn 1536 slots, 1130 live in 6 groups", 1536, 1536, True)]:
        buf = torch.from_numpy(rs.standard_normal((1, max(nq, nk), 1536)).astype(np.float32)).cuda()
        q, k, v = buf[:, :nq, :512], buf[:, :nk, 512:1024], buf[:, :nk, 1024:]
        cam = gs = None
        if grouped:
            bounds = [0, 190, 380, 570, 760, 950, 1130]
            cam = torch.full((nq,), -1, dtype=torch.int32)
            for c in range(6):
                cam[bounds[c]:bounds[c + 1]] = c
            cam, gs = cam.cuda(), torch.tensor(bounds, dtype=torch.int32).cuda()
        res = {}
        for split in (False, True):
            res[split] = timed(lambda: ops.attention_f32(q, k, v, 8, cam, gs, split=split))
        a, b = ops.attention_f32(q, k, v, 8, cam, gs, split=False), ops.attention_f32(q, k, v, 8, cam, gs, split=True)
        print(f"{name}: exact fp32 {res[False]:.1f} us, split f16 {res[True]:.1f} us; max |difference| {float((a - b).abs().max()):.2e}", flush=True)

    # the block a frame runs: projections + attention core + output product (layers.fused_graph_attention), three routes
    import torch.nn as nn
    from simpb_amd.plugin import routes  # noqa: E402
    from simpb_amd.plugin.layers import MultiheadAttention, fused_graph_attention  # noqa: E402
    torch.manual_seed(0)
    layer = MultiheadAttention(512, 8, batch_first=True).cuda().eval()
    pre, post = nn.Linear(256, 512, bias=False).cuda(), nn.Linear(512, 256, bias=False).cuda()
    f, p = torch.randn(1, 900, 256, device="cuda"), torch.randn(1, 900, 256, device="cuda")
    tf, tp = torch.randn(1, 600, 256, device="cuda"), torch.randn(1, 600, 256, device="cuda")
    with torch.no_grad():
        for name, (query, key, value, kp) in [("gnn block", (f, None, f, None)), ("temp_gnn block", (f, tf, tf, tp))]:
            line = []
            for label, sw in [("fp32 gemm + fp32 attention", dict(gemm_split_fp16=False, attention_split_fp16=False)),
                              ("split gemm + fp32 attention", dict(gemm_split_fp16=True, attention_split_fp16=False)),
                              ("split gemm + split-half attention", dict(gemm_split_fp16=True, attention_split_fp16=True))]:
                with routes.override(**sw):
                    line.append(f"{label} {timed(lambda: fused_graph_attention(layer, pre, post, query, p, key, kp, value)):.1f} us")
            print(f"{name}: " + "; ".join(line), flush=True)


if __name__ == "__main__":
    main()
