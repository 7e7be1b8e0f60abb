"""Does MIOpen's fused conv+bias+ReLU beat F.conv2d + the in-place bias_act kernel on the backbone's shapes?
usage: python tools/bench_conv_fused.py"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from simpb_amd.plugin.ops import bias_act_  # noqa: E402

torch.backends.cudnn.benchmark = True
SHAPES = [  # (cin, cout, k, stride, H, W) at 6 images
    (64, 64, 1, 1, 64, 176), (64, 64, 3, 1, 64, 176), (64, 256, 1, 1, 64, 176), (256, 128, 1, 1, 64, 176),
    (128, 128, 3, 2, 64, 176), (128, 512, 1, 1, 32, 88), (256, 256, 3, 1, 16, 44), (512, 2048, 1, 1, 8, 22),
]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * iters) * 1e3


for cin, cout, k, st, h, w in SHAPES:
    x = torch.randn(6, cin, h, w, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last) * 0.05
    b = torch.randn(cout, device="cuda", dtype=torch.half)
    pad = k // 2
    ours = lambda: bias_act_(F.conv2d(x, wt, None, st, pad), b, None, relu=True)  # noqa: E731
    try:
        fused = lambda: torch.ops.aten.miopen_convolution_relu(x, wt, b, [st, st], [pad, pad], [1, 1], 1)  # noqa: E731
        ref, got = ours(), fused()
        err = float((ref.float() - got.float()).abs().max())
        t_f = timeit(fused)
    except Exception as e:  # noqa: BLE001
        err, t_f = float("nan"), float("nan")
        print("fused failed:", str(e)[:100])
    print(f"{cin:4d}->{cout:4d} k{k} s{st} {h}x{w}: conv+bias_act {timeit(ours):6.1f} us   miopen fused {t_f:6.1f} us   max diff {err:.3g}")
