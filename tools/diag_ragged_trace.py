"""Where does a batch of independent streams leave the batch-of-one result, and why?

A batch of independent streams (FrameRunner(independent_streams=True): ONE flat 2D slot array over bs x 6 camera groups,
csrc/alloc.hip alloc_scatter_ragged_kernel) must return, stream by stream, what a batch-of-one runner returns. Round 3's
whole-stream comparison went red at (stream 1, frame 5) = the time-jump frame of the stream that jumps; this tool finds the
cause from that failure instead of re-running it: for every stream it

  1. runs the batch and, per stream, a plain batch-of-one runner on the SAME recorded features over the whole stream and
     reports the first frame whose detections differ beyond 1e-3 (as sets);
  2. at that frame records every operator boundary of the decoder (forward hooks on the head's layers + the bank calls) in
     three runs: B = the batch (the stream's own rows cut out of the flat layout), S = batch-of-one from the state the BATCH
     held for the stream before the frame (same state: a difference is this frame's arithmetic in the flat layout), O =
     batch-of-one from its OWN state (a difference S vs O is history: earlier ~1e-6 differences carried by the bank);
  3. prints, operator by operator, max |B - S| and max |S - O| and, at the first record that leaves 1e-4, what decided it:
     for a ranking (InstanceBank.update / cache, decoder top-k) the scores on either side of the cut with their gap, for
     the allocation the anchors whose inside / outside test flipped with their distance to the image border in pixels.

usage: python tools/diag_ragged_trace.py [--bs 3] [--frames 8] [--jump 1,5] [--wh 352 128] [--all]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.runner import FrameRunner  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=3)
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--jump", default="1,5")
ap.add_argument("--wh", type=int, nargs=2, default=(352, 128))
ap.add_argument("--all", action="store_true", help="trace every (stream, frame) that differs, not only the first per stream")
ap.add_argument("--tol", type=float, default=1e-3)
ap.add_argument("--trace", action="append", default=[], metavar="STREAM,FRAME", help="trace this (stream, frame) whether or not it differs")
args = ap.parse_args()
bs, frames, wh = args.bs, args.frames, tuple(args.wh)
jump = None if args.jump in ("", "none") else (int(args.jump.split(",")[0]), int(args.jump.split(",")[1]), 10.0)
CAP = 1536
forced = {(int(x.split(",")[0]), int(x.split(",")[1])) for x in args.trace}


def make():
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    return model.cuda().fuse_conv_bn().half_backbone()


class Replay(torch.nn.Module):
    """Serves recorded feature tensors instead of running a backbone."""

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


def one(m, b):
    return dict(projection_mat=m["projection_mat"][b:b + 1], image_wh=m["image_wh"][b:b + 1],
                timestamp=m["timestamp"][b:b + 1], img_metas=[m["img_metas"][b]])


# ---------------------------------------------------------------------------------------------- operator-boundary records
class Recorder:
    """Records every tensor an operator of the head returns, cut down to ONE stream's rows: 3D-state tensors [bs, 900, .]
    -> [b]; flat 2D-state tensors [1, bs * cap, .] -> the stream's live slots (from the allocation in force)."""

    def __init__(self, head, stream, nstreams):
        self.head, self.b, self.n = head, stream, nstreams
        self.items, self.order = {}, []
        self.alloc = None
        self.hooks = []
        self.saved = {}

    def _cut(self, t):
        if not torch.is_tensor(t) or t.dim() < 2:
            return None
        t = t.detach()
        if self.n > 1 and t.shape[0] == self.n:
            return t[self.b].float().cpu()
        if self.n > 1 and t.shape[0] == 1 and t.shape[1] == self.n * 900:   # a 3D-state tensor joined as one [1, bs * 900, .] view
            return t[0, self.b * 900:(self.b + 1) * 900].float().cpu()
        if self.alloc is not None and t.shape[0] == 1:
            gs = self.alloc["gs"]
            lo, hi = (int(gs[self.b * 6]), int(gs[(self.b + 1) * 6])) if self.n > 1 else (0, int(gs[6]))
            if t.shape[1] == self.alloc["slots"]:
                return t[0, lo:hi].float().cpu()
        if t.shape[0] == 1:
            return t[0].float().cpu()
        return None

    def add(self, name, value):
        from simpb_amd.plugin.dense import Segments
        vals = []

        def flat(x):
            if isinstance(x, Segments):
                vals.append(x.materialize())
            elif torch.is_tensor(x):
                vals.append(x)
            elif isinstance(x, (list, tuple)):
                for y in x:
                    flat(y)
        flat(value)
        for k, t in enumerate(vals):
            c = self._cut(t)
            if c is not None:
                key = f"{name}.{k}"
                while key in self.items:
                    key += "'"
                self.items[key] = c
                self.order.append(key)

    def attach(self):
        head = self.head
        for i, (op, layer) in enumerate(zip(head.operation_order, head.layers)):
            if layer is None:
                continue
            name = f"L{i:02d}.{op}"

            def hook(mod, inp, out, name=name, op=op):
                if op == "allocation":
                    a = mod.last
                    gs = a.group_start.detach().cpu().numpy()
                    self.alloc = dict(gs=gs, slots=a.q2a.shape[1], a=a)
                    lo, hi = (int(gs[self.b * 6]), int(gs[(self.b + 1) * 6])) if self.n > 1 else (0, int(gs[6]))
                    q2a = a.q2a[0, lo:hi].cpu() - (self.b * 900 if self.n > 1 else 0)
                    cam = a.query_cam[lo:hi].cpu() - (self.b * 6 if self.n > 1 else 0)
                    self.items[name + ".q2a"], self.items[name + ".cam"] = q2a.float(), cam.float()
                    self.items[name + ".is_center"] = a.is_center[0, lo:hi].float().cpu()
                    self.order += [name + ".q2a", name + ".cam", name + ".is_center"]
                    self.add(name + ".ref_pts2d", out[0])
                else:
                    self.add(name, out)
            self.hooks.append(layer.register_forward_hook(hook))
        bank = head.instance_bank
        for meth in ("get", "rank_current", "update", "cache_and_assign_ids"):
            orig = getattr(bank, meth)
            self.saved[meth] = orig

            def wrap(*a, _orig=orig, _m=meth, **kw):
                out = _orig(*a, **kw)
                if _m == "update":   # inputs too: the classification the ranking saw
                    self.add("bank.update.in_cls", a[2])
                if _m == "cache_and_assign_ids":
                    self.add("bank.cache.in_cls", a[2])
                    self.add("bank.cache.in_anchor", a[1])
                self.add("bank." + _m, out)
                return out
            setattr(bank, meth, wrap)
        for nm in ("anchor_encoder", "anchor_encoder2d"):
            self.hooks.append(getattr(head, nm).register_forward_hook(lambda m, i, o, nm=nm: self.add(nm, o)))
        return self

    def detach(self):
        for h in self.hooks:
            h.remove()
        for meth, orig in self.saved.items():
            setattr(self.head.instance_bank, meth, orig)


def set_state(bank, state, b=None, prev=None):
    for k, v in state.items():
        bank._static[k].copy_(v if (b is None or v.dim() == 0) else v[b:b + 1])
    bank.has_history = prev is not None
    bank.metas = prev


def diff(a, b):
    if a.shape != b.shape:
        return float("inf")
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def row_set_diff(a, b):
    """max over rows of a of the distance to the nearest row of b (inf norm): 0 when b holds the same rows in another order."""
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or not a.shape[0] or not b.shape[0]:
        return float("inf")
    return float(torch.cdist(a.double(), b.double(), p=float("inf")).min(dim=1).values.max())


def explain(name, rb, rs, tag):
    """What decided the first diverging record `name` between the runs rb and rs."""
    a, b = rb.items[name], rs.items[name]
    print(f"      [{tag}] first record beyond 1e-4: {name}  shapes {tuple(a.shape)} / {tuple(b.shape)}  max|d| {diff(a, b):.3e}  "
          f"as row sets {row_set_diff(a, b):.3e}", flush=True)
    if a.shape == b.shape and a.dim() == 2:
        d = (a.double() - b.double()).abs().max(dim=1).values
        bad = torch.nonzero(d > 1e-4).flatten()
        print(f"        rows beyond 1e-4: {len(bad)} of {a.shape[0]}; first {bad[:8].tolist()} with {[f'{float(x):.2e}' for x in d[bad[:8]]]}")
    if name.startswith("bank.update") or name.startswith("bank.rank_current"):
        for who, r in (("B", rb), ("S", rs)):
            cls = r.items.get("bank.update.in_cls.0")
            if cls is not None:
                s = cls.max(dim=-1).values
                v, _ = torch.sort(s, descending=True)
                k = 300
                print(f"        {who}: update ranks max-class logits; rank {k - 1} / {k}: {float(v[k - 1]):.9f} / {float(v[k]):.9f}  gap {float(v[k - 1] - v[k]):.3e}; "
                      f"smallest gap among ranks 290..310: {float((v[289:309] - v[290:310]).min()):.3e}")
    if name.startswith("bank.cache") or name.startswith("bank.get"):
        pass
    if ".allocation" in name:
        an, bn = rb.items[name.rsplit(".", 1)[0] + ".q2a"], rs.items[name.rsplit(".", 1)[0] + ".q2a"]
        ac, bc = rb.items[name.rsplit(".", 1)[0] + ".cam"], rs.items[name.rsplit(".", 1)[0] + ".cam"]
        sa = {(int(x), int(c)) for x, c in zip(an.tolist(), ac.tolist())}
        sb = {(int(x), int(c)) for x, c in zip(bn.tolist(), bc.tolist())}
        print(f"        (anchor, camera) pairs only in the first run: {sorted(sa - sb)[:10]}; only in the second: {sorted(sb - sa)[:10]}")


def border_report(anchor_b, anchor_s, metas_b, pairs, wh):
    """For (anchor, cam) pairs whose allocation flipped: the projected centre / corner pixels and their distance to the
    border in both runs (allocation.py:55-83: centre valid = inside the image; corner valid = depth > 0 and inside)."""
    proj = metas_b["projection_mat"][0].double().cpu()
    for a, c in pairs[:6]:
        for who, anc in (("B", anchor_b), ("S", anchor_s)):
            x = anc[a].double()
            ctr = torch.cat([x[:3], x.new_ones(1)])
            p = proj[c] @ ctr
            u, v = float(p[0] / p[2].clamp(min=1e-5)), float(p[1] / p[2].clamp(min=1e-5))
            du = min(u, wh[0] - u)
            dv = min(v, wh[1] - v)
            print(f"          {who}: anchor {a} cam {c}: centre pixel ({u:.5f}, {v:.5f}) depth {float(p[2]):.5f}; distance to the border: "
                  f"{min(du, dv):.3e} px")


def compare_traces(rb, rs, tag, stop_at=1e-4):
    first = None
    for name in rb.order:
        if name not in rs.items:
            continue
        a, b = rb.items[name], rs.items[name]
        d = diff(a, b)
        mark = ""
        if first is None and not d <= stop_at:
            first = name
            mark = "   <-- first beyond 1e-4"
        scale = float(b.abs().max()) if b.numel() else 0.0
        rs_ = row_set_diff(a, b) if (a.dim() == 2 and a.shape == b.shape and d > stop_at) else d
        nbad = int(((a.double() - b.double()).abs().max(dim=1).values > stop_at).sum()) if (a.dim() == 2 and a.shape == b.shape) else -1
        print(f"      [{tag}] {name:44s} {str(tuple(a.shape)):16s} max|d| {d:.3e}  as row sets {rs_:.3e}  rows beyond 1e-4: {nbad:5d}  (scale {scale:.2e}){mark}", flush=True)
    return first


def rows3(r):
    b = np.asarray(r["boxes_3d"], np.float64)
    return torch.as_tensor(np.concatenate([b[:, :6], np.sin(b[:, 6:7]), np.cos(b[:, 6:7]), b[:, 7:], np.asarray(r["scores_3d"], np.float64)[:, None],
                                           np.asarray(r["labels_3d"], np.float64)[:, None] * 10.0], axis=1))


def rows2(r):
    return torch.as_tensor(np.concatenate([np.asarray(r["boxes_2d"], np.float64) * 1e-2, np.asarray(r["scores_2d"], np.float64)[:, None],
                                           np.asarray(r["labels_2d"], np.float64)[:, None] * 10.0], axis=1))


def unmatched(have, want, tol):
    if have.shape[0] == 0 or want.shape[0] == 0:
        return max(have.shape[0], want.shape[0])
    return int((torch.cdist(want, have, p=float("inf")).min(dim=1).values > tol).sum())


# ---------------------------------------------------------------------------------------------- 1. the batch
imgs = [synth.images(bs, f % 4, wh).cuda() for f in range(frames)]
metas = [synth.frame_metas(bs, f, wh, jump=jump) for f in range(frames)]
model = make()
batch = FrameRunner(model, bs, (wh[1], wh[0]), capacity=CAP, device=torch.device("cuda"), use_graph=False, independent_streams=True)
inner, last = model.extract_feat, {}


def spy(img):
    last["fm"] = inner(img)
    return last["fm"]


model.extract_feat = spy
bank = batch.head.instance_bank
got, seen, states = [], [], []
for f in range(frames):
    torch.cuda.synchronize()
    states.append({k: v.clone() for k, v in bank._static.items()})
    got.append(batch.step(imgs[f], metas[f]))
    seen.append([t.clone() for t in list(last["fm"])[:3]])
print(f"batch of {bs} independent streams, {frames} frames, jump {jump}: {batch.stats}", flush=True)

# ---------------------------------------------------------------------------------------------- 2. per stream
for b in range(bs):
    replay = Replay(make().head)
    plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=CAP, device=torch.device("cuda"), use_graph=False)
    own_states = []
    first_bad = None
    for f in range(frames):
        torch.cuda.synchronize()
        own_states.append({k: v.clone() for k, v in plain.head.instance_bank._static.items()})
        replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
        want = plain.step(plain.img, one(metas[f], b))[0]["img_bbox"]
        have = got[f][b]["img_bbox"]
        u3, u2 = unmatched(rows3(have), rows3(want), args.tol), unmatched(rows2(have), rows2(want), args.tol)
        sd = {k: diff(states[f][k][b:b + 1] if states[f][k].dim() else states[f][k], own_states[f][k]) for k in ("cached_anchor", "cached_feature", "confidence")}
        print(f"stream {b} frame {f}: unmatched 3D rows {u3}/{len(want['boxes_3d'])}, 2D rows {u2}/{len(want['boxes_2d'])} (have {len(have['boxes_2d'])}); "
              f"state in front of the frame, batch vs own: " + ", ".join(f"{k} {v:.2e}" for k, v in sd.items()), flush=True)
        differs = bool(u3 or u2 or len(have["boxes_2d"]) != len(want["boxes_2d"]))
        if (differs and (first_bad is None or args.all)) or (b, f) in forced:
            first_bad = f if (first_bad is None and differs) else first_bad
            # ---- 3. the three traced runs of this frame
            prev = dict(img_metas=[metas[f - 1]["img_metas"][b]]) if f else None
            prev_b = dict(img_metas=metas[f - 1]["img_metas"]) if f else None
            keep_own = {k: v.clone() for k, v in plain.head.instance_bank._static.items()}
            keep_b = {k: v.clone() for k, v in bank._static.items()}
            keep_prev_plain, keep_prev_b = plain.prev_metas, batch.prev_metas

            def traced(runner, state, sb, prev_m, n, img, m):
                set_state(runner.head.instance_bank, state, sb, prev_m)
                if prev_m is None:
                    runner.head.instance_bank.reset()
                runner.prev_metas = prev_m
                rec = Recorder(runner.head, b if n > 1 else 0, n).attach()
                try:
                    out = runner.step(img, m)
                finally:
                    rec.detach()
                torch.cuda.synchronize()
                for k, v in runner.head.instance_bank._static.items():   # the state the frame leaves behind
                    if v.dim() >= 2:
                        t = v[rec.b] if n > 1 else v[0]
                        rec.items["state_after." + k] = (t if t.dim() == 2 else t[:, None]).float().cpu()
                        rec.order.append("state_after." + k)
                return rec, out

            model.extract_feat = lambda img: seen[f]
            rB, _ = traced(batch, states[f], None, prev_b, bs, imgs[f], metas[f])
            model.extract_feat = spy
            replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
            rS, oS = traced(plain, states[f], b, prev, 1, plain.img, one(metas[f], b))
            rO, oO = traced(plain, own_states[f], None, prev, 1, plain.img, one(metas[f], b))
            wS = oS[0]["img_bbox"]
            print(f"   traced: batch vs same-state batch-of-one: unmatched 3D {unmatched(rows3(have), rows3(wS), args.tol)}, 2D "
                  f"{unmatched(rows2(have), rows2(wS), args.tol)}; same-state vs own-state: 3D {unmatched(rows3(wS), rows3(want), args.tol)}", flush=True)
            for tag, x, y in (("B vs S: flat layout, same state", rB, rS), ("S vs O: same code, own history", rS, rO)):
                print(f"   --- {tag}")
                firstrec = compare_traces(x, y, tag[:6])
                if firstrec is not None:
                    explain(firstrec, x, y, tag[:6])
                    if ".allocation" in firstrec:
                        pre = firstrec.rsplit(".", 1)[0]
                        sa = {(int(q), int(c)) for q, c in zip(x.items[pre + ".q2a"].tolist(), x.items[pre + ".cam"].tolist())}
                        sb_ = {(int(q), int(c)) for q, c in zip(y.items[pre + ".q2a"].tolist(), y.items[pre + ".cam"].tolist())}
                        flipped = sorted(sa ^ sb_)
                        # the anchors the allocation saw: the last 3D anchor record in front of it
                        idx = x.order.index(pre + ".q2a")
                        anc = [n for n in x.order[:idx] if x.items[n].dim() == 2 and x.items[n].shape[-1] == 11]
                        if anc and flipped:
                            border_report(x.items[anc[-1]], y.items[anc[-1]], one(metas[f], b), flipped, wh)
            # put both runners back where the whole-stream pass left them
            set_state(plain.head.instance_bank, keep_own, None, plain.prev_metas)
            plain.prev_metas = dict(img_metas=[metas[f]["img_metas"][b]])
            plain.head.instance_bank.metas = plain.prev_metas
            set_state(bank, keep_b, None, keep_prev_b)
            batch.prev_metas = keep_prev_b
            # the whole-stream pass of `plain` must continue from ITS state after frame f: re-run the frame untraced
            set_state(plain.head.instance_bank, own_states[f], None, prev)
            if prev is None:
                plain.head.instance_bank.reset()
            plain.prev_metas = prev
            replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
            plain.step(plain.img, one(metas[f], b))
    print(f"stream {b}: first differing frame {first_bad}", flush=True)


# ---------------------------------------------------------------------------------------------- 4. how fast do two fp32 runs drift apart?
# Two batch-of-one runs of the SAME code on the SAME features, the second with its bank features scaled by (1 + 1e-6 * noise)
# in front of frame 1 -- the size of a summation-order difference. If their distance grows per frame like the distance
# between the batch and the batch-of-one run above, the growth is the decoder's (random weights), not the flat layout's.
def state_distance(sa, sb_):
    out = {}
    for k in ("cached_feature", "cached_anchor"):
        a, b_ = sa[k][0].float().cpu(), sb_[k][0].float().cpu()
        out[k] = row_set_diff(a, b_)
    out["confidence"] = float((torch.sort(sa["confidence"][0].cpu()).values - torch.sort(sb_["confidence"][0].cpu()).values).abs().max())
    return out


print("\n--- drift of two batch-of-one runs whose bank features differ by 1e-6 (relative) in front of frame 1, as row sets", flush=True)
for b in range(bs):
    runs = []
    for perturb in (False, True):
        replay = Replay(make().head)
        plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=CAP, device=torch.device("cuda"), use_graph=False)
        sts, res = [], []
        for f in range(frames):
            torch.cuda.synchronize()
            if perturb and f == 1:
                g = torch.Generator(device="cuda").manual_seed(7)
                cf = plain.head.instance_bank._static["cached_feature"]
                cf.mul_(1.0 + 1e-6 * torch.randn(cf.shape, device="cuda", generator=g))
            sts.append({k: v.clone() for k, v in plain.head.instance_bank._static.items()})
            replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
            res.append(plain.step(plain.img, one(metas[f], b))[0]["img_bbox"])
        runs.append((sts, res))
    for f in range(frames):
        d = state_distance(runs[0][0][f], runs[1][0][f])
        db = state_distance({k: (v[b:b + 1] if v.dim() else v) for k, v in states[f].items()}, runs[0][0][f])
        u = unmatched(rows3(runs[1][1][f]), rows3(runs[0][1][f]), args.tol)
        print(f"stream {b} frame {f}: perturbed vs plain: state " + ", ".join(f"{k} {v:.2e}" for k, v in d.items()) + f"; unmatched 3D rows {u}/300"
              f"   |   batch vs plain: state " + ", ".join(f"{k} {v:.2e}" for k, v in db.items()), flush=True)
