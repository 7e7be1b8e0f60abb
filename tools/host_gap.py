"""Where a pipelined step's wall time goes on the HOST: launch() (staging + graph launches), the wait for the decoder
enqueued one step earlier, and the numpy finish. Since the decoder of a frame is enqueued in the step that feeds it, the
decoder stream has its next job while the host does all this.
usage: python tools/host_gap.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.plugin.detection3d import SparseBox3DDecoder  # noqa: E402
from simpb_amd.runner import PipelinedRunner, SplitPipelinedRunner  # noqa: E402

wh = (704, 256)
cfg = configs.simpb_plus(anchor=synth.anchors(900))
model = plugin.build_detector(cfg["model"]).eval()
synth.load_procedural(model)
model = model.cuda().fuse_conv_bn().half_backbone()
torch.backends.cudnn.benchmark = True
r = (SplitPipelinedRunner if "--split" in sys.argv else PipelinedRunner)(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"))
imgs = [synth.images(1, f, wh).cuda() for f in range(4)]
metas = [synth.frame_metas(1, f, wh) for f in range(140)]
for f in range(20):
    r.step(imgs[f % 4], metas[f])
torch.cuda.synchronize()
orig = SparseBox3DDecoder.decode_static_host
acc = dict(launch=0.0, wait=0.0, finish=0.0, decode=0.0)


def timed_decode(*a, **k):
    t = time.perf_counter()
    out = orig(*a, **k)
    acc["decode"] += time.perf_counter() - t
    return out


SparseBox3DDecoder.decode_static_host = staticmethod(timed_decode)
n = 100
t0 = time.perf_counter()
for f in range(20, 20 + n):
    a = time.perf_counter()
    r.launch(imgs[f % 4], metas[f])       # backbone(f) and decoder(f) enqueued; decoder(f-1) is what collect() waits for
    b = time.perf_counter()
    r.queue[0]["done"].synchronize()
    c = time.perf_counter()
    r.collect()
    d = time.perf_counter()
    acc["launch"] += b - a
    acc["wait"] += c - b
    acc["finish"] += d - c
total = time.perf_counter() - t0
print(f"{n} steps: {total / n * 1e3:.3f} ms/step; host launch() {acc['launch'] / n * 1e3:.3f} ms, wait for the GPU {acc['wait'] / n * 1e3:.3f} ms, "
      f"collect() after the wait {acc['finish'] / n * 1e3:.3f} ms (of which numpy finish {acc['decode'] / n * 1e3:.3f} ms)")
