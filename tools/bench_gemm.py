"""Standalone timing of csrc/gemm.hip on the decoder's shapes vs the vendor GEMM route
(torch.cat + F.linear). usage: python tools/bench_gemm.py [iters]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from simpb_amd.plugin import dense  # noqa: E402

SHAPES = [  # (name, M, N, segment widths, relu)
    ("qkv 3D", 900, 1536, [256, 256], False),
    ("q 3D", 900, 512, [256, 256], False),
    ("kv temp", 600, 1024, [256, 256], False),
    ("attn out fold", 900, 256, [512, 256, 256], False),
    ("ffn fc1", 900, 1024, [512], True),
    ("ffn out fold", 900, 256, [1024, 512], False),
    ("dfa logits", 900, 416, [256, 256], False),
    ("output_proj", 900, 256, [256], False),
    ("qkv 2D cap", 1536, 1536, [256, 256], False),
    ("msda in", 1536, 384, [256, 256], False),
]


def timeit(fn, iters, reps=5):
    """us per call with `iters` calls captured in one hipGraph (host launch cost out of the picture)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
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
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":  # a few eager launches per shape, for rocprofv3 --pmc
        for name, m, n, ks, relu in SHAPES:
            xs = [torch.randn(m, k, device="cuda") for k in ks]
            w = torch.randn(n, sum(ks), device="cuda") / sum(ks) ** 0.5
            b = torch.randn(n, device="cuda")
            for _ in range(3):
                dense.linear(xs, w, b, relu=relu)
        torch.cuda.synchronize()
        return
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for name, m, n, ks, relu in SHAPES:
        xs = [torch.randn(m, k, device="cuda") for k in ks]
        w = torch.randn(n, sum(ks), device="cuda") / sum(ks) ** 0.5
        b = torch.randn(n, device="cuda")
        out = torch.empty(m, n, device="cuda")
        t_ours = timeit(lambda: dense.linear(xs, w, b, relu=relu, out=out), iters)
        t_vendor = timeit(lambda: F.linear(torch.cat(xs, 1) if len(xs) > 1 else xs[0], w, b), iters)

        def cold():
            flush.zero_()
            dense.linear(xs, w, b, relu=relu, out=out)
        t_flush = timeit(lambda: flush.zero_(), 10, 2)
        t_cold = timeit(cold, 10, 2) - t_flush
        gflop = 2 * m * n * sum(ks) / 1e9
        print(f"{name:14s} M={m:5d} N={n:5d} K={sum(ks):5d}  ours {t_ours:6.1f} us ({gflop / t_ours * 1e3:6.1f} TF/s)  "
              f"cold {t_cold:6.1f} us  vendor cat+linear {t_vendor:6.1f} us")


if __name__ == "__main__":
    main()
