"""Error of the grouped GEMM against float64, exact-fp32 kernel vs the split-f16 (four passes) kernel, on the decoder's shapes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import dense, routes  # noqa: E402

for m, n, ks in [(900, 1536, [256, 256]), (900, 256, [512, 256, 256]), (900, 1024, [512]), (900, 256, [1024, 512]), (1130, 256, [2176]), (1536, 384, [256, 256])]:
    rs = torch.Generator().manual_seed(m + n)
    xs = [torch.randn(m, k, generator=rs) for k in ks]
    w = torch.randn(n, sum(ks), generator=rs) / np.sqrt(sum(ks))
    b = torch.randn(n, generator=rs)
    want = torch.cat(xs, 1).double() @ w.double().t() + b.double()
    out = {}
    for split in (False, True):
        with routes.override(gemm_split_fp16=split):
            got = dense.linear([x.cuda() for x in xs], w.cuda(), b.cuda()).cpu().double()
        e = (got - want).abs()
        out[split] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    print(f"M {m} N {n} K {sum(ks)}: fp32 kernel max {out[False][0]:.2e} rms {out[False][1]:.2e} | split-f16 x4 max {out[True][0]:.2e} rms {out[True][1]:.2e} | "
          f"ratio max {out[True][0] / out[False][0]:.2f} rms {out[True][1] / out[False][1]:.2f}  (|y| max {float(want.abs().max()):.2f})", flush=True)
