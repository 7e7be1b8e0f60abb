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
    ap.add_argument("--dummy-backbone", action="store_true",
                    help="backbone stream runs big matmuls + a copy of pre-computed features instead of the convolutions")
    ap.add_argument("--no-miopen", action="store_true", help="torch.backends.cudnn.enabled = False: PyTorch's own convolutions")
    ap.add_argument("--trace", type=int, default=-1, help="frame whose decoder inputs / per-layer outputs are compared")
    args = ap.parse_args()
    from simpb_amd import configs, plugin, synth
    from simpb_amd.runner import FrameRunner, PipelinedRunner
    wh = (352, 128)
    dev = torch.device("cuda")
    if args.no_serialize:
        PipelinedRunner.SERIALIZE_EAGER = False
    if args.no_miopen:
        torch.backends.cudnn.enabled = False
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

    def spy_on(runner, store):
        """Clone (on the decoder's stream, no host sync) what the decoder of every frame is given and returns."""
        head, inner = runner.head, runner.head.forward
        bank = head.instance_bank

        def flat(prefix, v, out):
            if torch.is_tensor(v):
                out[prefix] = v.detach().clone()
            elif isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    flat(f"{prefix}.{i}", x, out)
            elif isinstance(v, dict):
                for k, x in v.items():
                    flat(f"{prefix}.{k}", x, out)

        cur = {"rec": None, "seq": 0}

        def hook_for(name):
            def hook(mod, inp, out):
                if cur["rec"] is not None:
                    flat(f"mod.{cur['seq']:04d}.{name}", out, cur["rec"])
                    cur["seq"] += 1
            return hook

        for name, mod in head.named_modules():
            if name:
                mod.register_forward_hook(hook_for(name))

        inner_get = bank.get

        def spy_get(*a, **k):
            if cur["rec"] is not None and k.get("dn_metas") is None and len(a) >= 2 and "bank_inputs" in a[1]:
                flat("bank_get.in.T_dt", list(a[1]["bank_inputs"]), cur["rec"])
                st2 = getattr(bank, "_static", None) or {}
                if "cached_anchor" in st2:
                    flat("bank_get.in.stored", st2["cached_anchor"], cur["rec"])
            res = inner_get(*a, **k)
            if cur["rec"] is not None:
                flat("bank_get.out", [x for x in res], cur["rec"])
            return res

        bank.get = spy_get

        def forward(fm, metas, *a, **k):
            rec = {}
            cur["rec"], cur["seq"] = rec, 0
            flat("in.fm", list(fm)[:3], rec)
            flat("in.values", list(fm)[3] if len(fm) > 3 else None, rec)
            flat("in.proj", metas.get("projection_mat"), rec)
            flat("in.bank_inputs", metas.get("bank_inputs"), rec)
            st = getattr(bank, "_static", None) or {}
            flat("in.bank", {k2: v for k2, v in st.items() if torch.is_tensor(v)}, rec)
            outs = inner(fm, metas, *a, **k)
            flat("out", {k2: v for k2, v in outs.items() if k2 != "alloc_list"}, rec)
            store.append(rec)
            return outs

        head.forward = forward

    frames = args.frames
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]
    bad = 0
    for rep in range(args.reps):
        r = PipelinedRunner(make(), 1, (wh[1], wh[0]), capacity=1536, device=dev, use_graph=not args.eager)
        if args.one_stream:
            r.s_bb = r.s_head
        if args.dummy_backbone:
            model = r.model
            with torch.no_grad():
                pre = [[t.clone() for t in model.extract_feat(imgs[f])] for f in range(frames)]
            torch.cuda.synchronize()
            load_a = torch.randn(4096, 4096, device=dev)
            state = {"f": 0}

            def fake_features(slot, pre=pre, state=state, load_a=load_a, r=r):
                for _ in range(12):
                    load_a @ load_a
                fm = [t.clone() for t in pre[state["f"] % frames]]
                state["f"] += 1
                fm.append(r.head.precompute_values(fm))
                return fm

            r._features = fake_features
        seen, out = [], []
        pipe_trace, plain_trace = [], []
        if args.trace >= 0:
            spy_on(r, pipe_trace)
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
            if args.trace >= 0 and trial == 0:
                spy_on(plain, plain_trace)
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
        if args.trace >= 0 and max(float(x) for x in lines[0].split()) > 1e-3:
            first_bad = next(i for i, x in enumerate(lines[0].split()) if float(x) > 1e-3)
            args.trace = first_bad
            a, b = pipe_trace[args.trace], plain_trace[args.trace]
            for key in a:
                if key in b and a[key].shape == b[key].shape:
                    d = float((a[key].double() - b[key].double()).abs().max()) if a[key].numel() else 0.0
                    if d != 0.0 or key.startswith("in.") or key.startswith("bank_get"):
                        print(f"    trace frame {args.trace} {key:40s} max|pipe - plain| = {d:.3e}  max|pipe| = {float(a[key].double().abs().max()) if a[key].numel() else 0:.3e}")
                else:
                    print(f"    trace frame {args.trace} {key:40s} only in pipe or shape differs")
        print(f"rep {rep}: pipe-vs-plainA {lines[0]}\n        pipe-vs-plainB {lines[1]}", flush=True)
    print(f"BAD {bad}/{args.reps}", flush=True)


if __name__ == "__main__":
    main()
