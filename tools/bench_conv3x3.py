"""csrc/conv3x3.hip, every tiling, against the vendor convolution + the in-place bias_act pass on the 3x3 shapes of
ResNet50 + FPN at 6 x 256 x 704 (and R101 1408 x 512 with --big). Rotates 4 input buffers so that the activations do not
stay in L2 between repetitions. usage: python tools/bench_conv3x3.py [--big]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import simpb_amd  # noqa: E402,F401  (vendor solver settings)
from simpb_amd.plugin.ops import bias_act_, conv3x3_nhwc  # noqa: E402

torch.backends.cudnn.benchmark = True


def timeit(fn, reps=30):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


big = "--big" in sys.argv
H0, W0 = (128, 352) if big else (64, 176)
SHAPES = [("layer1 conv2", 64, 64, H0, W0, 1), ("layer2.0 conv2", 128, 128, H0, W0, 2), ("layer2 conv2", 128, 128, H0 // 2, W0 // 2, 1),
          ("layer3.0 conv2", 256, 256, H0 // 2, W0 // 2, 2), ("layer3 conv2", 256, 256, H0 // 4, W0 // 4, 1),
          ("layer4.0 conv2", 512, 512, H0 // 4, W0 // 4, 2), ("layer4 conv2", 512, 512, H0 // 8, W0 // 8, 1),
          ("fpn 0", 256, 256, H0, W0, 1), ("fpn 1", 256, 256, H0 // 2, W0 // 2, 1), ("fpn 2", 256, 256, H0 // 4, W0 // 4, 1),
          ("fpn 3", 256, 256, H0 // 8, W0 // 8, 1)]
total = {"vendor": 0.0, "auto": 0.0, "best": 0.0}
COUNT = {"layer1 conv2": 3, "layer2 conv2": 3, "layer3 conv2": 5, "layer4 conv2": 2}
for name, cin, cout, h, w, stride in SHAPES:
    xs = [torch.randn(6, cin, h, w, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.half) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda", dtype=torch.half)
    vendor = lambda i: bias_act_(F.conv2d(xs[i & 3], wt, None, stride, 1), b, None, relu=True)  # noqa: E731
    ref = F.conv2d(xs[0].float(), wt.float(), b.float(), stride, 1).relu()
    ho, wo = ref.shape[2:]
    gflop = 2 * 6 * ho * wo * cout * 9 * cin / 1e9
    tv = timeit(vendor)
    line = f"{name:15s} {cin:3d}->{cout:3d} {h:3d}x{w:3d} s{stride} {gflop:5.2f} GF  vendor+bias_act {tv:6.1f} us |"
    best = 1e9
    for v in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        fn = lambda i: conv3x3_nhwc(xs[i & 3], wt, b, True, stride, variant=v)  # noqa: E731
        err = float((fn(0).float() - ref).abs().max())
        t = timeit(fn)
        if v:
            best = min(best, t)
        else:
            t0 = t
        line += f" v{v} {t:6.1f} us ({gflop / t * 1e-3:4.0f} TF/s, err {err:.1e})" if v == 0 else f" v{v} {t:6.1f}"
    n = COUNT.get(name, 1)
    total["vendor"] += n * tv
    total["auto"] += n * t0
    total["best"] += n * best
    print(line, flush=True)
print("per frame (R50 layer counts):", {k: round(v, 1) for k, v in total.items()}, "us")
