"""Is a difference between the plain and the pipelined runner born in the vendor backbone or in the decoder?

Runs the whole detector (fp32 ResNet50+FPN through MIOpen + decoder) over the same 12 frames with the eager, the
graph and the pipelined runner (the latter several times), and prints per frame whether the backbone feature maps
are bit-identical to the eager run's and how far the detections are from it. Bit-identical features with different
detections would be a race in the runner / decoder; different features are the vendor's convolutions.

    python tools/pipe_determinism.py [--reps 4] [--fp16]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def checksum(tensors):
    return [int(t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32).long().sum()) for t in tensors[:3]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--fp16", action="store_true")
    args = ap.parse_args()
    from simpb_amd import configs, plugin, synth
    from simpb_amd.runner import FrameRunner, PipelinedRunner
    wh = (352, 128)

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        model = model.cuda().fuse_conv_bn()
        if args.fp16:
            model.half_backbone()
        return model

    frames = args.frames
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]

    def run(kind):
        model = make()
        sums = []
        if kind == "pipe":
            r = PipelinedRunner(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True)
            out = []
            for f in range(frames):
                out.append(r.step(imgs[f], metas[f]))
                if f >= 1:
                    sums.append(checksum(r.fm[(f - 1) % 2]))
            r.s_bb.synchronize()   # step() does not wait for the backbone of the frame just fed
            sums.append(checksum(r.fm[(frames - 1) % 2]))
            out = out[1:] + [r.flush()]
        else:
            r = FrameRunner(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=kind == "graph")
            inner = model.extract_feat
            last = {}

            def spy(img):
                last["fm"] = inner(img)
                return last["fm"]

            model.extract_feat = spy
            out = []
            for f in range(frames):
                out.append(r.step(imgs[f], metas[f]))
                sums.append(checksum(list(last["fm"])))
        return sums, [o[0]["img_bbox"] for o in out]

    base_sums, base = run("eager")
    report = []
    for kind in ["eager", "graph"] + ["pipe"] * args.reps:
        sums, det = run(kind)
        rows = []
        for f in range(frames):
            same_fm = sums[f] == base_sums[f]
            ds = float((det[f]["scores_3d"] - base[f]["scores_3d"]).abs().max())
            db = float((det[f]["boxes_3d"] - base[f]["boxes_3d"]).abs().max())
            rows.append((f, same_fm, ds, db))
        report.append({"kind": kind, "frames": rows})
        print(kind, " ".join(f"{f}:{'=' if s else 'X'}{ds:.0e}" for f, s, ds, db in rows), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pipe_determinism.json", "w") as fh:
        json.dump(report, fh)


if __name__ == "__main__":
    main()
