"""What bounds a pipelined frame: the replayed backbone graph alone, the replayed decoder graph alone, and both side by
side (the steady state of runner.PipelinedRunner), timed with events on their streams. usage: python tools/stream_times.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.runner import PipelinedRunner  # noqa: E402

wh = (704, 256)
cfg = configs.simpb_plus(anchor=synth.anchors(900))
model = plugin.build_detector(cfg["model"]).eval()
synth.load_procedural(model)
model = model.cuda().fuse_conv_bn().half_backbone()
torch.backends.cudnn.benchmark = True
r = PipelinedRunner(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"))
imgs = [synth.images(1, f, wh).cuda() for f in range(4)]
metas = [synth.frame_metas(1, f, wh) for f in range(60)]
for f in range(24):
    r.step(imgs[f % 4], metas[f])
torch.cuda.synchronize()
assert all(g is not None for g in r.bb_graph) and all(g is not None for g in r.head_graph), "graphs not captured yet"
n = 50


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def bb():
    with torch.cuda.stream(r.s_bb):
        for i in range(n):
            r.bb_graph[i & 1].replay()


def dec():
    with torch.cuda.stream(r.s_head):
        for i in range(n):
            r.head_graph[i & 1].replay()


def both():
    for i in range(n):
        with torch.cuda.stream(r.s_bb):
            r.bb_graph[i & 1].replay()
        with torch.cuda.stream(r.s_head):
            r.head_graph[i & 1].replay()


for name, fn in (("backbone graph alone", bb), ("decoder graph alone", dec), ("both streams side by side", both)):
    fn()
    print(f"{name:28s} {wall(fn):.3f} ms per frame", flush=True)
