"""Does the decoder give the same detections when it runs beside the backbone as when it runs alone?

Per repetition: the pipelined runner over 12 frames of the whole detector, recording the feature maps each decoder
read; the same maps are then served to the plain eager runner. Prints the first frame (if any) whose detections
differ by more than 1e-3 and the largest difference. Variants:
    default      two streams, graphs
    --eager      two streams, no graphs
    --one-stream graphs, backbone and decoder on ONE stream (no concurrency)

    python tools/pipe_race.py --reps 6 [--eager | --one-stream] [--load]
`--load` first churns the allocator the way the test-suite does before the runner tests (blocks with old contents).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--one-stream", action="store_true")
    ap.add_argument("--load", action="store_true")
    ap.add_argument("--no-serialize", action="store_true", help="PipelinedRunner.SERIALIZE_EAGER = False")
    args = ap.parse_args()
    from simpb_amd import configs, plugin, synth
    from simpb_amd.runner import FrameRunner, PipelinedRunner
    wh = (352, 128)
    dev = torch.device("cuda")
    if args.no_serialize:
        PipelinedRunner.SERIALIZE_EAGER = False
    if args.load:
        junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 20, 1 << 22, 1 << 24, 3 << 20, 5 << 18) for _ in range(8)]
        del junk

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn()

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

    frames = args.frames
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]
    bad = 0
    for rep in range(args.reps):
        r = PipelinedRunner(make(), 1, (wh[1], wh[0]), capacity=1536, device=dev, use_graph=not args.eager)
        if args.one_stream:
            r.s_bb = r.s_head
        seen, out = [], []
        for f in range(frames):
            out.append(r.step(imgs[f], metas[f], force_eager=args.eager))
            if f >= 1:
                seen.append([t.clone() for t in r.fm[(f - 1) % 2][:3]])
        seen.append([t.clone() for t in r.fm[(frames - 1) % 2][:3]])
        out = out[1:] + [r.flush()]
        lines = []
        for trial in range(2):
            replay = Replay(make().head)
            plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=1536, device=dev, use_graph=False)
            diffs = []
            for f in range(frames):
                replay.load(seen[f])
                a = plain.step(plain.img, metas[f])[0]["img_bbox"]
                b = out[f][0]["img_bbox"]
                diffs.append(max(float((a["scores_3d"] - b["scores_3d"]).abs().max()),
                                 float((a["boxes_3d"][:, :6] - b["boxes_3d"][:, :6]).abs().max())))
            lines.append(" ".join(f"{d:.0e}" for d in diffs))
            if trial == 0 and max(diffs) > 1e-3:
                bad += 1
        print(f"rep {rep}: pipe-vs-plainA {lines[0]}\n        pipe-vs-plainB {lines[1]}", flush=True)
    print(f"BAD {bad}/{args.reps}", flush=True)


if __name__ == "__main__":
    main()
