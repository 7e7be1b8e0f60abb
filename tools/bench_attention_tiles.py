"""Per-key-tile cost of the packed-operand attention kernels: 900 queries against 900 / 1800 / 3600 / 7200 keys (the fixed
costs -- launch, Q, the meeting of the waves -- cancel in the differences)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd.plugin import ops  # noqa: E402
from tools.bench_attention import timed  # noqa: E402


def pack(x):
    hi = x.half()
    lo = ((x - hi.float()) * 2048.0).half()
    return ((hi.view(torch.int16).to(torch.int32) & 0xFFFF) | (lo.view(torch.int16).to(torch.int32) << 16)).view(torch.float32)


rs = np.random.RandomState(0)
prev = None
for nk in (900, 1800, 3600, 7200):
    buf = torch.from_numpy(rs.standard_normal((1, nk, 1536)).astype(np.float32)).cuda()
    p = pack(buf)
    line = [f"Nk {nk}:"]
    t_exact = timed(lambda: ops.attention_f32(buf[:, :900, :512], buf[:, :, 512:1024], buf[:, :, 1024:], 8, split=0), 100)
    t_pack = timed(lambda: ops.attention_f32(p[:, :900, :512], p[:, :, 512:1024], p[:, :, 1024:], 8, split=2), 100)
    line.append(f"exact fp32 {t_exact:.1f} us, pre-split halfs, eight waves {t_pack:.1f} us")
    if prev is not None:
        dt = (nk - prev[0]) / 32
        line.append(f"per 32-key tile step of all workgroups: exact {(t_exact - prev[1]) / dt * 1e3:.0f} ns, packed {(t_pack - prev[2]) / dt * 1e3:.0f} ns")
    prev = (nk, t_exact, t_pack)
    print(" ".join(line), flush=True)
