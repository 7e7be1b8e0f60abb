"""Grouped split-operand GEMM: time against K at fixed M, N (fixed cost per launch vs cost per 64-deep chunk)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import dense  # noqa: E402
from tools.bench_attention import timed  # noqa: E402

for m, n in [(900, 1536), (900, 256), (1130, 1024)]:
    prev = None
    for k in (128, 256, 512, 1024, 2048):
        x = torch.randn(1, m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / 20
        b = torch.randn(n, device="cuda")
        t = timed(lambda: dense.linear(x, w, b), 100)
        line = f"M {m} N {n} K {k}: {t:.1f} us"
        if prev is not None:
            line += f"   per additional 64-deep chunk: {(t - prev[1]) / ((k - prev[0]) / 64) * 1e3:.0f} ns"
        prev = (k, t)
        print(line, flush=True)
