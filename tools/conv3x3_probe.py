"""A few launches of csrc/conv3x3.hip tilings on the largest shapes, for rocprofv3 --pmc passes.
usage: python tools/conv3x3_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from simpb_amd.plugin.ops import conv3x3_nhwc  # noqa: E402

CASES = [("fpn0", 256, 256, 64, 176, (5, 6)), ("layer1", 64, 64, 64, 176, (1, 5)), ("layer3", 256, 256, 16, 44, (1, 5))]
for name, cin, cout, h, w, variants in CASES:
    x = torch.randn(6, cin, h, w, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.half) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", dtype=torch.half)
    for v in variants:
        for _ in range(6):
            conv3x3_nhwc(x, wt, b, True, 1, variant=v)
torch.cuda.synchronize()
