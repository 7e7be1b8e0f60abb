"""value_proj (group_attn.py:176) at 89 760 tokens x 256 x 256: three-pass split kernel on f32 tokens against the two-pass
kernel on the f16 tokens the FPN leaves (csrc/linear_split.hip). usage: python tools/bench_value_proj.py"""
import sys

import torch

sys.path.insert(0, ".")
from simpb_amd.plugin.ops import linear_f32, linear_split  # noqa: E402


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


m = 89760
xs = [torch.randn(m, 256, device="cuda").half() for _ in range(3)]
xf = [x.float() for x in xs]
w = torch.randn(256, 256, device="cuda") / 16
b = torch.randn(256, device="cuda")
print(f"three passes, f32 tokens: {timeit(lambda i: linear_split(xf[i % 3], w, b)):6.1f} us")
print(f"two passes,   f16 tokens: {timeit(lambda i: linear_split(xs[i % 3], w, b)):6.1f} us")
print(f"exact fp32 matrix cores:  {timeit(lambda i: linear_f32(xf[i % 3], w, b)):6.1f} us")
