"""Standalone timing (hipGraph-captured) of the fused MLP-chain launches of the decoder.
usage: python tools/bench_chain.py"""
import sys

import torch

sys.path.insert(0, ".")
from simpb_amd.plugin.detection2d import SparseBox2DEncoder, SparseBox2DRefinementModule  # noqa: E402
from simpb_amd.plugin.detection3d import SparseBox3DEncoder, SparseBox3DRefinementModule  # noqa: E402


def timeit(fn, iters=30, reps=5):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * reps) * 1e3


def main():
    import os
    if len(sys.argv) > 1:  # a variant library (experiments)
        import simpb_amd._lib as L
        L.LIB = os.path.abspath(sys.argv[1])
    torch.manual_seed(0)
    enc3 = SparseBox3DEncoder([128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4).cuda().eval()
    enc2 = SparseBox2DEncoder(256, with_sin_embed=True, in_loops=1, out_loops=2).cuda().eval()
    r3 = SparseBox3DRefinementModule(256, num_cls=10, refine_yaw=True, with_quality_estimation=True).cuda().eval()
    r2 = SparseBox2DRefinementModule(256, num_cls=10, with_alpha_branch=True).cuda().eval()
    a3 = torch.randn(1, 900, 11, device="cuda")
    f3, e3 = torch.randn(1, 900, 256, device="cuda"), torch.randn(1, 900, 256, device="cuda")
    a2 = torch.rand(1, 1536, 2, device="cuda")
    f2, e2 = torch.randn(1, 1536, 256, device="cuda"), torch.randn(1, 1536, 256, device="cuda")
    dt = torch.tensor([0.5], device="cuda")
    for name, fn in [
        ("anchor encoder 3D (900 rows, 4 chains x 4 stages)", lambda: enc3(a3)),
        ("refine3d reg only (900)", lambda: r3(f3, a3, e3, time_interval=dt, return_cls=False)),
        ("refine3d reg+cls+quality (900)", lambda: r3(f3, a3, e3, time_interval=dt, return_cls=True)),
        ("encoder 2D sine (1536)", lambda: enc2(a2)),
        ("refine2d reg+cls+alpha (1536)", lambda: r2(f2, a2, e2)),
    ]:
        from simpb_amd.plugin import routes
        row = []
        for r4 in (True, False):
            with routes.override(chain_rows4=r4):
                row.append(timeit(fn))
        print(f"{name:52s} rows4 {row[0]:7.1f} us   rows16 {row[1]:7.1f} us")


if __name__ == "__main__":
    main()
