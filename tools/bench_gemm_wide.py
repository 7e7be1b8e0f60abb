"""Grouped split-operand GEMM: narrow tiles (32 x 64, K split over the waves) vs wide tiles (64 x 128, one tile per wave),
the decoder's shapes at one stream and at eight (the library switches to the wide form at 400 tiles of 64 x 128; the numbers under
profiles/r04_gemm_wide_tiles.txt were taken with that threshold moved by a measurement build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import dense  # noqa: E402
from tools.bench_attention import timed  # noqa: E402

for name, m, n, ks in [("q|k|v 900 x 512 -> 1536", 900, 1536, [256, 256]), ("q|k|v N2 1536 rows (1130 live)", 1536, 1536, [256, 256]),
                       ("ffn fc1 900 x 512 -> 1024", 900, 1024, [512]), ("ffn fc1 1130 x 512 -> 1024", 1130, 1024, [512]),
                       ("ffn out 900 x 1536 -> 256", 900, 256, [1024, 512]), ("msda 1130 x 2176 -> 256", 1130, 256, [2176 - 128, 128]),
                       ("bs8 q|k|v 7200 x 512 -> 1536", 7200, 1536, [256, 256]), ("bs8 ffn fc1 7200 -> 1024", 7200, 1024, [512]),
                       ("bs8 ffn out 7200 x 1536 -> 256", 7200, 256, [1024, 512]), ("bs8 attn out 7200 x 1024 -> 256", 7200, 256, [512, 256, 256])]:
    xs = [torch.randn(1, m, k, device="cuda") for k in ks]
    w = torch.randn(n, sum(ks), device="cuda") / 20
    b = torch.randn(n, device="cuda")
    t = timed(lambda: dense.linear(xs, w, b), 100)
    gf = 2.0 * m * n * sum(ks) / 1e9
    print(f"{name}: {t:.1f} us  ({gf / t * 1e3:.0f} TFLOP/s fp32-equivalent)", flush=True)
