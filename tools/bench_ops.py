"""Micro-benchmark of the two samplers at the shipped R50 704x256 shapes with a realistic
sampling-location distribution (synthetic 6-camera ring). Prints one JSON line per
kernel with the algorithmic bytes of SURVEY.md §8(d) and the achieved rate. GPU only."""
import json
import math
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import synth  # noqa: E402
from simpb_amd.plugin import ops  # noqa: E402


def realistic_daf_inputs(bs=1, image_wh=(704, 256), device="cuda", seed=0):
    anchors = torch.from_numpy(synth.anchors(900, seed=seed))[None].repeat(bs, 1, 1)
    rs = np.random.RandomState(seed)
    fix = torch.tensor([[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]])
    learn = torch.from_numpy(rs.uniform(-0.5, 0.5, (bs, 900, 6, 3)).astype(np.float32))
    size = anchors[..., None, 3:6].exp()
    kp = torch.cat([fix * size, learn * size], dim=-2)
    c, s = anchors[..., 7], anchors[..., 6]
    x = c[..., None] * kp[..., 0] - s[..., None] * kp[..., 1]
    y = s[..., None] * kp[..., 0] + c[..., None] * kp[..., 1]
    kp = torch.stack([x, y, kp[..., 2]], -1) + anchors[..., None, :3]
    proj = torch.from_numpy(synth.camera_rig(image_wh))
    ext = torch.cat([kp, torch.ones_like(kp[..., :1])], -1)
    pts = torch.einsum("kij,bapj->bapki", proj, ext)
    loc = pts[..., :2] / pts[..., 2:3].clamp(min=1e-5) / torch.tensor(image_wh, dtype=torch.float32)
    w = torch.from_numpy(rs.uniform(0, 1, (bs, 900, 13, 6, 4, 8)).astype(np.float32))
    w = w / w.sum(dim=(2, 3, 4), keepdim=True)
    return loc.contiguous().to(device), w.to(device)


def time_kernel(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters * 1e-3


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    shapes = synth.level_shapes()
    tokens = 6 * sum(h * w for h, w in shapes)
    feat = torch.randn(bs, tokens, 256, device="cuda")
    ss = torch.tensor([shapes] * 6, dtype=torch.int32, device="cuda")
    sizes = [h * w for h, w in shapes] * 6
    ssi = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]).reshape(6, 4), dtype=torch.int32, device="cuda")
    loc, w = realistic_daf_inputs(bs)
    valid = int(((loc > 0) & (loc < 1)).all(-1).sum())
    t = time_kernel(lambda: ops.deformable_aggregation_function(feat, ss, ssi, loc, w))
    # V*L*4 taps*C*4B + loc + weights + out
    nbytes = valid * 4 * 4 * 256 * 4 + loc.numel() * 4 + w.numel() * 4 + bs * 900 * 256 * 4
    print(json.dumps(dict(kernel="daf_fwd_rows", bs=bs, valid_triples=valid, frac_valid=valid / (bs * 900 * 78),
                          us=t * 1e6, algorithmic_MB=nbytes / 1e6, GBps=nbytes / t / 1e9)))

    nq = 1130
    value = torch.randn(bs, 6, tokens // 6, 8, 32, device="cuda")
    ss2 = torch.tensor(shapes, dtype=torch.long, device="cuda")
    lsi = torch.cat([ss2.new_zeros(1), ss2.prod(1).cumsum(0)[:-1]])
    ref = torch.rand(bs, nq, 1, 1, 1, 2, device="cuda")
    off = torch.randn(bs, nq, 8, 4, 4, 2, device="cuda") * 2.0
    sloc = ref + off / torch.stack([ss2[:, 1], ss2[:, 0]], -1)[None, None, None, :, None, :]
    aw = torch.rand(bs, nq, 8, 16, device="cuda").softmax(-1).view(bs, nq, 8, 4, 4)
    bounds = [int(round(nq * k / 6)) for k in range(7)]
    qcam = ops.query_cam_from_groups(list(zip(bounds[:-1], bounds[1:])), nq, "cuda")
    t = time_kernel(lambda: ops.ms_deform_attn_grouped(value, ss2, lsi, sloc, aw, qcam))
    nbytes = bs * nq * (65536 + 1536 + 1024)
    print(json.dumps(dict(kernel="msda_grouped_fwd", bs=bs, num_query=nq, us=t * 1e6, algorithmic_MB=nbytes / 1e6,
                          GBps=nbytes / t / 1e9)))


if __name__ == "__main__":
    main()


def bench_linear():
    import torch.nn.functional as F
    for (m, n, k) in [(89760, 256, 256), (359040, 256, 256), (1536, 1024, 512), (900, 256, 256)]:
        x = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        b = torch.randn(n, device="cuda")
        t1 = time_kernel(lambda: ops.linear_f32(x, w, b), iters=20)
        t2 = time_kernel(lambda: F.linear(x, w, b), iters=20)
        fl = 2.0 * m * n * k
        print(json.dumps(dict(kernel="linear_f32_mfma", m=m, n=n, k=k, us=t1 * 1e6, tflops=fl / t1 / 1e12,
                              vendor_us=t2 * 1e6, vendor_tflops=fl / t2 / 1e12)))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "linear":
    bench_linear()
