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
    ap.add_argument("--no-serialize", action="store_true", help="kept for old command lines: the runner no longer serialises eager steps")
    ap.add_argument("--dummy-backbone", action="store_true",
                    help="backbone stream runs big matmuls + a copy of pre-computed features instead of the convolutions")
    ap.add_argument("--no-miopen", action="store_true", help="torch.backends.cudnn.enabled = False: PyTorch's own convolutions")
    ap.add_argument("--trace", type=int, default=-1, help="frame whose decoder inputs / per-layer outputs are compared")
    ap.add_argument("--late-clone", action="store_true", help="keep bank_get's output alive and clone it again at the end of the decoder")
    ap.add_argument("--equal-priority", action="store_true", help="both streams at the default priority")
    ap.add_argument("--verbose", action="store_true", help="with --trace: print every differing record of the first bad frame")
    ap.add_argument("--pre-kernel", action="store_true", help="one throw-away launch on the decoder stream in front of bank_get")
    ap.add_argument("--bank-diag", action="store_true", help="library built with -DSIMPB_BANK_DIAG: print bank_get's self-check log")
    ap.add_argument("--streams", type=int, default=1, help="camera streams (runners) launched side by side; the first is checked")
    ap.add_argument("--cu-split", type=int, default=0, help="decoder stream on every N-th CU, backbone stream on the others (disjoint CU masks)")
    ap.add_argument("--dummy-kind", default="matmul", help="with --dummy-backbone: matmul | conv3x3 | conv1x1 | value_proj | format | copy | none")
    ap.add_argument("--dump", default="", help="with --trace: save the bank_get operands of the first bad frame here and stop")
    args = ap.parse_args()
    from simpb_amd import configs, plugin, synth
    from simpb_amd.runner import FrameRunner, PipelinedRunner
    wh = (352, 128)
    dev = torch.device("cuda")
    if args.equal_priority:
        PipelinedRunner.STREAM_PRIORITIES = (0, 0)
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

        from simpb_amd.plugin import blocks as _blocks
        if not hasattr(_blocks, "_race_daf_orig"):
            _blocks._race_daf_orig = _blocks.DAF
            _blocks._race_sinks = []

            def daf_spy(feat, ss, ssi, loc, w):
                out = _blocks._race_daf_orig(feat, ss, ssi, loc, w)
                for sink, stream in _blocks._race_sinks:
                    if sink["rec"] is not None and torch.cuda.current_stream() == stream():
                        k = sink["daf"] = sink.get("daf", 0) + 1
                        flat(f"daf.{k:02d}.loc", loc, sink["rec"])
                        flat(f"daf.{k:02d}.weights", w, sink["rec"])
                        flat(f"daf.{k:02d}.out", out, sink["rec"])
                return out

            _blocks.DAF = daf_spy
        _blocks._race_sinks.append((cur, (lambda: runner.s_head) if hasattr(runner, "s_head") else torch.cuda.current_stream))
        inner_get = bank.get

        def spy_get(*a, **k):
            if cur["rec"] is not None and k.get("dn_metas") is None and len(a) >= 2 and "bank_inputs" in a[1]:
                flat("bank_get.in.T_dt", list(a[1]["bank_inputs"]), cur["rec"])
                st2 = getattr(bank, "_static", None) or {}
                if "cached_anchor" in st2:
                    flat("bank_get.in.stored", st2["cached_anchor"], cur["rec"])
            if args.pre_kernel and getattr(bank, "_static", None):
                bank._static["prev_id"].add_(0)  # a throw-away launch between the staged copies and bank_get
            res = inner_get(*a, **k)
            if cur["rec"] is not None:
                flat("bank_get.out", [x for x in res], cur["rec"])
                if args.late_clone:
                    cur["warped"] = res[3]
            return res

        bank.get = spy_get

        def forward(fm, metas, *a, **k):
            if torch.cuda.is_current_stream_capturing():  # a capture is not a frame: no record, and no clones in the graph
                cur["rec"] = None
                return inner(fm, metas, *a, **k)
            rec = {}
            cur["rec"], cur["seq"], cur["daf"] = rec, 0, 0
            flat("in.fm", list(fm)[:3], rec)
            flat("in.values", list(fm)[3] if len(fm) > 3 else None, rec)
            flat("in.proj", metas.get("projection_mat"), rec)
            flat("in.bank_inputs", metas.get("bank_inputs"), rec)
            st = getattr(bank, "_static", None) or {}
            flat("in.bank", {k2: v for k2, v in st.items() if torch.is_tensor(v)}, rec)
            outs = inner(fm, metas, *a, **k)
            flat("out", {k2: v for k2, v in outs.items() if k2 != "alloc_list"}, rec)
            if cur.get("warped") is not None:
                flat("bank_get.out3_late", cur["warped"], rec)  # same tensor, cloned again at the end of the decoder
            store.append(rec)
            return outs

        head.forward = forward

    frames = args.frames
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]
    bad = 0
    for rep in range(args.reps):
        r = PipelinedRunner(make(), 1, (wh[1], wh[0]), capacity=1536, device=dev, use_graph=not args.eager)
        if args.cu_split:
            from _streams import cu_masked_streams
            r.s_bb, r.s_head = cu_masked_streams(dev, args.cu_split)
        if args.one_stream:
            r.s_bb = r.s_head
        if args.dummy_backbone:
            model = r.model
            with torch.no_grad():
                pre = [[t.clone() for t in model.extract_feat(imgs[f])] for f in range(frames)]
            torch.cuda.synchronize()
            load_a = torch.randn(4096, 4096, device=dev)
            state = {"f": 0}

            import torch.nn.functional as F
            from simpb_amd.plugin import ops as _ops
            kind = args.dummy_kind
            cx = torch.randn(6, 256, 64, 176, device=dev).half().contiguous(memory_format=torch.channels_last)
            cw3 = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).half().contiguous(memory_format=torch.channels_last)
            cw1 = (torch.randn(256, 256, 1, 1, device=dev) * 0.02).half()
            cb = torch.zeros(256, device=dev).half()
            lv = [torch.randn(6, 256, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
                  for h, w in ((32, 88), (16, 44), (8, 22), (4, 11))] if wh == (352, 128) else None
            big = torch.randn(23 * 1024 * 1024, device=dev)

            def load(pre_fm):
                """One kernel type on the backbone stream, sized to last about as long as a decoder."""
                if kind == "matmul":
                    for _ in range(12):
                        load_a @ load_a
                elif kind == "conv3x3":
                    for _ in range(24):
                        F.conv2d(cx, cw3, None, padding=1)
                elif kind == "conv1x1":
                    for _ in range(60):
                        _ops.conv1x1_nhwc(cx, cw1, cb, None, True, 1)
                elif kind == "value_proj":
                    for _ in range(6):
                        r.head.precompute_values(pre_fm)
                elif kind == "format":
                    for _ in range(40):
                        _ops.format_tokens(lv, 1, 6)
                elif kind == "copy":
                    for _ in range(30):
                        big.clone()
                elif kind != "none":
                    raise SystemExit(f"unknown --dummy-kind {kind}")

            def fake_features(slot, pre=pre, state=state, load_a=load_a, r=r):
                load(pre[state["f"] % frames])
                src = pre[state["f"] % frames]
                fm = [src[0].clone(), pre[0][1], pre[0][2]]  # ONE pair of (H, W, start) tables (validated once, on the host)
                state["f"] += 1
                fm.append(r.head.precompute_values(fm))
                return fm

            r._features = fake_features
        seen, out = [], []
        pipe_trace, plain_trace = [], []
        from simpb_amd.plugin import blocks as _b
        if hasattr(_b, "_race_sinks"):
            _b._race_sinks.clear()  # records of earlier repetitions
        if args.trace >= 0:
            spy_on(r, pipe_trace)
        others = [PipelinedRunner(make(), 1, (wh[1], wh[0]), capacity=1536, device=dev, use_graph=not args.eager)
                  for _ in range(args.streams - 1)]  # further camera streams launched beside the traced one (bench --streams)
        for f in range(frames):
            r.launch(imgs[f], metas[f], force_eager=args.eager)
            for k, o in enumerate(others):
                if f - (k + 1) >= 0:  # each a frame behind the previous one
                    o.launch(imgs[f - (k + 1)], metas[f - (k + 1)], force_eager=args.eager)
            out.append(r.collect())
            for k, o in enumerate(others):
                if f - (k + 1) >= 0:
                    o.collect()
            if f >= 1:
                seen.append([t.clone() for t in r.fm[(f - 1) % 2][:3]])
        r.s_bb.synchronize()   # collect() does not wait for the backbone of the frame just fed
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
            first_bad = next(i for i, x in enumerate(lines[0].split()) if float(x) > 0)  # bit-exact otherwise
            args.trace = first_bad
            if args.trace >= len(pipe_trace):
                print(f"    first differing frame {args.trace} was a graph replay (no per-module records)", flush=True)
                print(f"rep {rep}: pipe-vs-plainA {lines[0]}", flush=True)
                continue
            a, b = pipe_trace[args.trace], plain_trace[args.trace]
            k3 = "bank_get.out.3"
            if k3 in a and k3 in b:
                pa, pb = a[k3][0].cpu(), b[k3][0].cpu()
                rows = (pa != pb).any(1).nonzero().flatten().tolist()
                cols = (pa != pb).any(0).nonzero().flatten().tolist()
                Tm = a["bank_get.in.T_dt.0"][0].cpu()
                vals = sorted(set(round(float(x), 5) for x in pa[rows][:, cols].flatten().tolist()))[:6]
                print(f"    FAULT frame {args.trace}: rows {rows[:3]}..{rows[-3:]} (n={len(rows)}) cols {cols} bad values {vals} "
                      f"T row0 {[round(float(x), 5) for x in Tm[0]]} row1 {[round(float(x), 5) for x in Tm[1]]} row2 {[round(float(x), 5) for x in Tm[2]]}", flush=True)
            first = [key for key in a if key in b and a[key].shape == b[key].shape and a[key].numel()
                     and not torch.equal(a[key], b[key])][:6]
            print(f"    first differing records of frame {args.trace}: {first}", flush=True)
            if first:
                x, y = a[first[0]].cpu().float(), b[first[0]].cpu().float()
                x2, y2 = x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])
                rows = (x2 != y2).any(1).nonzero().flatten().tolist()
                cols = (x2 != y2).any(0).nonzero().flatten().tolist()
                print(f"    {first[0]} shape {tuple(x.shape)}: {len(rows)} rows differ {rows[:8]}..{rows[-3:]}, {len(cols)} cols "
                      f"{cols[:6]}..{cols[-3:]}, max|diff| {float((x2 - y2).abs().max()):.3e}, pipe there "
                      f"{[round(float(v), 4) for v in x2[rows[0], cols[:6]]]} plain {[round(float(v), 4) for v in y2[rows[0], cols[:6]]]}", flush=True)
            if first and first[0].startswith("daf.") and first[0].endswith(".out"):
                # is the wrong segment an OLD value of the same addresses (a lost store / stale line) or a new wrong value?
                x, y = a[first[0]].cpu()[0], b[first[0]].cpu()[0]
                rows = (x != y).any(1).nonzero().flatten().tolist()
                cols = (x[rows[0]] != y[rows[0]]).nonzero().flatten()
                seg = x[rows[0]][cols]
                cands = []
                for fi in range(max(0, args.trace - 2), args.trace + 1):
                    for k2, v2 in pipe_trace[fi].items():
                        if k2.startswith("daf.") and k2.endswith(".out") and not (fi == args.trace and k2 == first[0]):
                            cands.append((fi, k2, v2.cpu()[0]))
                hits = [(fi, k2) for fi, k2, v2 in cands if torch.equal(v2[rows[0]][cols], seg)]
                near = sorted(((float((v2[rows[0]][cols] - seg).abs().max()), fi, k2) for fi, k2, v2 in cands))[:3]
                print(f"    wrong segment row {rows[0]} cols {int(cols[0])}..{int(cols[-1])}: identical to an earlier DAF output at the same "
                      f"place: {hits}; nearest earlier outputs (max|diff|, frame, record): {near}; |pipe-plain| there "
                      f"{float((seg - y[rows[0]][cols]).abs().max()):.3e}", flush=True)
            for key in (a if args.verbose else ()):
                if key in b and a[key].shape == b[key].shape:
                    d = float((a[key].double() - b[key].double()).abs().max()) if a[key].numel() else 0.0
                    if d != 0.0 or key.startswith("in.") or key.startswith("bank_get"):
                        print(f"    trace frame {args.trace} {key:40s} max|pipe - plain| = {d:.3e}  max|pipe| = {float(a[key].double().abs().max()) if a[key].numel() else 0:.3e}")
                else:
                    print(f"    trace frame {args.trace} {key:40s} only in pipe or shape differs")
            if args.dump:
                keep = lambda tr, f: {k: v.cpu() for k, v in tr[f].items() if k.startswith(("bank_get", "in.bank"))}  # noqa: E731
                torch.save({"frame": first_bad, "pipe": keep(pipe_trace, first_bad), "plain": keep(plain_trace, first_bad),
                            "pipe_prev": keep(pipe_trace, first_bad - 1), "plain_prev": keep(plain_trace, first_bad - 1)},
                           args.dump)
                print(f"rep {rep}: dumped frame {first_bad} to {args.dump}", flush=True)
                bad_stop = True
        if args.bank_diag:
            import ctypes
            import numpy as np
            from simpb_amd import _lib
            fn = _lib.lib().simpb_debug_bank_faults
            fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int
            cnt = ctypes.c_uint(0)
            buf = np.zeros((64, 24), np.float32)
            torch.cuda.synchronize()
            fn(ctypes.byref(cnt), buf.ctypes.data)
            seen_cnt = getattr(args, "_diag_seen", 0)
            if cnt.value > seen_cnt:
                names = "i r0 q0 w0 r6 q6 w6 m0 m1 m2 m3 n0 n1 n2 n3 cx cy cz x2 y2 s c".split()
                for e in buf[seen_cnt:min(cnt.value, 64)]:
                    hw = e[22:24].view(np.uint32)
                    print("    DIAG " + " ".join(f"{k}={v:.5g}" for k, v in zip(names, e[:22])) + f" hw_id={hw[0]:#x} xcc={hw[1]:#x}", flush=True)
                args._diag_seen = cnt.value
        print(f"rep {rep}: pipe-vs-plainA {lines[0]}\n        pipe-vs-plainB {lines[1]}", flush=True)
        if args.dump and locals().get("bad_stop"):
            break
    print(f"BAD {bad}/{rep + 1}", flush=True)


if __name__ == "__main__":
    main()
