"""Batch of independent streams (FrameRunner, recorded features) vs a plain eager runner of batch one per stream.
usage: python tools/diag_ragged.py [bs] [--graph] [--jump] [--refbatch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.runner import FrameRunner, PipelinedRunner  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
graph = "--graph" in sys.argv
jump = None
if "--jump" in sys.argv:   # --jump [stream,frame]
    i = sys.argv.index("--jump")
    sf = sys.argv[i + 1].split(",") if i + 1 < len(sys.argv) and "," in sys.argv[i + 1] else ("1", "5")
    jump = (int(sf[0]), int(sf[1]), 10.0)
wh = (352, 128)


def make():
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    return model.cuda().fuse_conv_bn().half_backbone()


class Replay(torch.nn.Module):
    def __init__(self, head):
        super().__init__()
        self.head, self.maps = head, None

    def load(self, fm):
        if self.maps is None:
            self.maps = [t.clone() for t in fm]
        else:
            for d, s in zip(self.maps, fm):
                d.copy_(s)

    def extract_feat(self, img):
        return self.maps


frames = 8
imgs = [synth.images(bs, f % 4, wh).cuda() for f in range(frames)]
metas = [synth.frame_metas(bs, f, wh, jump=jump) for f in range(frames)]


def one(m, b):
    return dict(projection_mat=m["projection_mat"][b:b + 1], image_wh=m["image_wh"][b:b + 1],
                timestamp=m["timestamp"][b:b + 1], img_metas=[m["img_metas"][b]])


model = make()
got, seen = [], []
if "--pipe" in sys.argv:
    batch = PipelinedRunner(model, bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=graph,
                            independent_streams="--refbatch" not in sys.argv)
    for f in range(frames):
        got.append(batch.step(imgs[f], metas[f]))
        batch.s_bb.synchronize()
        if f >= 1:
            seen.append([t.clone() for t in list(batch.fm[(f - 1) % 2])[:3]])
    seen.append([t.clone() for t in list(batch.fm[(frames - 1) % 2])[:3]])
    got = got[1:] + [batch.flush()]
else:
    batch = FrameRunner(model, bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=graph,
                        independent_streams="--refbatch" not in sys.argv)
    inner, last = model.extract_feat, {}

    def spy(img):
        last["fm"] = inner(img)
        return last["fm"]

    model.extract_feat = spy
    for f in range(frames):
        got.append(batch.step(imgs[f], metas[f]))
        seen.append([t.clone() for t in list(last["fm"])[:3]])
print(batch.stats, flush=True)
for b in range(bs):
    replay = Replay(make().head)
    plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=False)
    for f in range(frames):
        replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
        w = plain.step(plain.img, one(metas[f], b))[0]["img_bbox"]
        g = got[f][b]["img_bbox"]
        ds = float(np.abs(np.sort(g["scores_3d"].numpy()) - np.sort(w["scores_3d"].numpy())).max())
        ga = torch.cat([g["boxes_3d"][:, :6], g["boxes_3d"][:, 7:], g["scores_3d"][:, None]], 1).double()
        wa = torch.cat([w["boxes_3d"][:, :6], w["boxes_3d"][:, 7:], w["scores_3d"][:, None]], 1).double()
        val, idx = torch.cdist(wa, ga, p=float("inf")).min(dim=1)
        frac = float((val <= 1e-3).double().mean())
        if ds > 1e-4 or "-v" in sys.argv:
            print(f"   matched rows within 1e-3: {frac:.3f}", end="")
            print(f"stream {b} frame {f}: scores(sorted) {ds:.2e} n2d {len(g['boxes_2d'])} vs {len(w['boxes_2d'])}", flush=True)
