"""Determinism soak of a batch of independent streams: the same frames through two fresh PipelinedRunners (replayed graphs, two
streams) must give bit-identical detections, frame by frame. usage: python tools/soak_independent.py [bs] [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.runner import PipelinedRunner  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
wh = (704, 256)


def make():
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    return model.cuda().fuse_conv_bn().half_backbone()


imgs = [synth.images(bs, f, wh).cuda() for f in range(4)]
metas = [synth.frame_metas(bs, f, wh) for f in range(frames)]
runs = []
for _ in range(2):
    r = PipelinedRunner(make(), bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), independent_streams=True)
    out = [r.step(imgs[f % 4], metas[f]) for f in range(frames)]
    runs.append(out[1:] + [r.flush()])
    print(r.stats, flush=True)
bad = 0
for f in range(frames):
    for b in range(bs):
        x, y = runs[0][f][b]["img_bbox"], runs[1][f][b]["img_bbox"]
        for k in ("boxes_3d", "scores_3d", "labels_3d", "boxes_2d", "scores_2d", "instance_ids"):
            if not np.array_equal(np.asarray(x[k]), np.asarray(y[k])):
                bad += 1
                print("differs:", f, b, k, flush=True)
                break
print(f"{frames} frames x {bs} streams: {bad} (frame, stream) pairs differ")
sys.exit(1 if bad else 0)
