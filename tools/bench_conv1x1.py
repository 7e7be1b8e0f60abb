"""csrc/conv1x1.hip (round-1 kernel) against the staged pipeline of csrc/conv3x3.hip (128 x 64 and 128 x 128 tiles) on the
1x1 shapes of ResNet50 + FPN at 6 x 256 x 704. Four rotating input buffers. usage: python tools/bench_conv1x1.py"""
import sys

import torch

sys.path.insert(0, ".")
from simpb_amd.plugin.ops import conv1x1_nhwc  # noqa: E402
from tools.bench_conv3x3 import timeit  # noqa: E402

# (name, cin, cout, h, w, stride, residual, count per frame)
SHAPES = [("l1.0 conv1", 64, 64, 64, 176, 1, False, 1), ("l1 conv3", 64, 256, 64, 176, 1, True, 3), ("l1.0 down", 64, 256, 64, 176, 1, False, 1),
          ("l1 conv1", 256, 64, 64, 176, 1, False, 2), ("l2.0 conv1", 256, 128, 64, 176, 1, False, 1), ("l2 conv3", 128, 512, 32, 88, 1, True, 4),
          ("l2.0 down", 256, 512, 64, 176, 2, False, 1), ("l2 conv1", 512, 128, 32, 88, 1, False, 3), ("l3.0 conv1", 512, 256, 32, 88, 1, False, 1),
          ("l3 conv3", 256, 1024, 16, 44, 1, True, 6), ("l3.0 down", 512, 1024, 32, 88, 2, False, 1), ("l3 conv1", 1024, 256, 16, 44, 1, False, 5),
          ("l4.0 conv1", 1024, 512, 16, 44, 1, False, 1), ("l4 conv3", 512, 2048, 8, 22, 1, True, 3), ("l4.0 down", 1024, 2048, 16, 44, 2, False, 1),
          ("l4 conv1", 2048, 512, 8, 22, 1, False, 2), ("fpn lat0", 256, 256, 64, 176, 1, True, 1), ("fpn lat1", 512, 256, 32, 88, 1, True, 1),
          ("fpn lat2", 1024, 256, 16, 44, 1, True, 1), ("fpn lat3", 2048, 256, 8, 22, 1, False, 1)]
tot = [0.0, 0.0, 0.0, 0.0, 0.0]
for name, cin, cout, h, w, stride, res, count in SHAPES:
    xs = [torch.randn(6, cin, h, w, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    wt = torch.randn(cout, cin, 1, 1, device="cuda", dtype=torch.half) * 0.05
    b = torch.randn(cout, device="cuda", dtype=torch.half)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.randn(6, cout, ho, wo, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last) if res else None
    ts = []
    for v in (0, 1, 2, 3):
        ts.append(timeit(lambda i: conv1x1_nhwc(xs[i & 3], wt, b, r, True, stride, variant=v)))
    mb = (6 * h * w * cin / (stride * stride) + 6 * ho * wo * cout * (2 if res else 1)) * 2 / 1e6
    print(f"{name:11s} {cin:4d}->{cout:4d} {h:2d}x{w:3d} s{stride} x{count}  {mb:6.1f} MB  auto {ts[0]:6.1f}  round-1 {ts[1]:6.1f}  128x64 {ts[2]:6.1f}  128x128 {ts[3]:6.1f} us"
          f"   best {mb / min(ts[1:]) * 1e-3:5.2f} TB/s", flush=True)
    for i in range(4):
        tot[i] += count * ts[i]
    tot[4] += count * min(ts[1:])
print("per frame: auto %.0f  round-1 %.0f  128x64 %.0f  128x128 %.0f  best %.0f us" % tuple(tot))
