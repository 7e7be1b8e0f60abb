"""Shared helpers for the parity tests (no reference code, no GPU needed)."""
import os

import numpy as np
import torch

from simpb_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_params(g):
    """Procedural parameters for the reference state_dict recorded in a head golden."""
    params = {}
    for key, shp in zip(g["state_keys"].tolist(), g["state_shapes"].tolist()):
        shape = tuple(int(s) for s in shp.split(",")) if shp else ()
        if key.endswith("fix_scale"):
            val = np.asarray([[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0],
                              [0, 0, 0.45], [0, 0, -0.45]], np.float32)  # config :224-232
        else:
            val = synth.procedural_tensor(key, shape)
        params[key] = torch.as_tensor(np.asarray(val))
    return params


def spec_of(g):
    s = {k[5:]: g[k] for k in g.files if k.startswith("spec_")}
    jump = tuple(s["jump"].tolist()) if s["jump"].ndim else None
    if jump is not None:
        jump = (int(jump[0]), int(jump[1]), float(jump[2]))
    return dict(image_wh=tuple(int(v) for v in s["image_wh"]), num_anchor=int(s["num_anchor"]),
                num_temp=int(s["num_temp"]), num_output=int(s["num_output"]), bs=int(s["bs"]),
                frames=int(s["frames"]), jump=jump, trace_frames=tuple(int(v) for v in s["trace_frames"]))


def rows_match(a, b, tol):
    """True when the rows of a are the rows of b up to order (each row of b has a distinct
    nearest row of a within tol). Instance order inside the temporal bank is decided by top-k over
    confidences that can sit 1e-7 apart, so two devices may legitimately hold the same set of
    instances in a different order ('Top-k tie order'); everything downstream is
    permutation-equivariant."""
    a = torch.as_tensor(np.asarray(a, np.float64)).reshape(-1, a.shape[-1])
    b = torch.as_tensor(np.asarray(b, np.float64)).reshape(-1, b.shape[-1])
    if a.shape != b.shape:
        return False
    d = torch.cdist(b, a, p=float("inf")) if a.shape[0] <= 4096 else None
    if d is None:
        return False
    val, idx = d.min(dim=1)
    return bool((val <= tol).all()) and len(torch.unique(idx)) == a.shape[0]


def compare_trace(got, g, prefix, rtol=2e-4, atol=2e-4, skip=(), allow_permutation=False):
    """Compare a synth.Trace against the golden arrays under `prefix`; returns the list of names
    checked. Float records use |a-b| <= atol + rtol*max|b| (sketches mix a row, so the scale of
    the whole record is the right yardstick); integer records must match exactly. With
    allow_permutation a record may also match as a SET of rows (see rows_match); integer records
    that index instances are then compared as multisets. `skip`: regular expressions of record
    names that are not compared (values a fused block never materialises)."""
    import re
    names = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    assert names, prefix
    skipped = lambda n: any(re.search(s, n) for s in skip)  # noqa: E731
    missing = [n for n in names if n not in got.items and not skipped(n)]
    assert not missing, f"trace records missing: {missing[:5]}"
    bad = []
    for n in names:
        if skipped(n):
            continue
        a, b = np.asarray(got.items[n]), g[prefix + n]
        if a.shape != b.shape:
            bad.append((n, "shape", a.shape, b.shape))
            continue
        if np.issubdtype(b.dtype, np.floating):
            tol = atol + rtol * float(np.abs(b).max() if b.size else 0.0)
            err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if b.size else 0.0
            if not err <= tol:
                if allow_permutation and b.ndim >= 2 and rows_match(a, b, tol):
                    continue
                bad.append((n, "err", err, tol))
        elif not np.array_equal(a.astype(np.int64), b.astype(np.int64)):
            if allow_permutation and ".q2a" in n:
                continue  # slot -> anchor index: anchor numbering itself is permuted
            if allow_permutation and np.array_equal(np.sort(a.reshape(-1)), np.sort(b.reshape(-1))):
                continue
            bad.append((n, "int mismatch", int((a != b).sum()), a.size))
    assert not bad, f"{len(bad)} of {len(names)} trace records differ, first: {bad[:6]}"
    return names


def _match_rows(got, want, col_tol):
    """Permutation p with got[p[i]] ~ want[i] (every column within its tolerance), or None."""
    got = torch.as_tensor(np.asarray(got, np.float64)) / torch.as_tensor(col_tol, dtype=torch.float64)
    want = torch.as_tensor(np.asarray(want, np.float64)) / torch.as_tensor(col_tol, dtype=torch.float64)
    if got.shape != want.shape:
        return None
    if got.shape[0] == 0:
        return np.zeros(0, np.int64)
    val, idx = torch.cdist(want, got, p=float("inf")).min(dim=1)
    if not bool((val <= 1.0).all()) or len(torch.unique(idx)) != got.shape[0]:
        return None
    return idx.numpy()


def compare_result(res, g, prefix, box_tol=1e-3, score_tol=1e-3):
    """One sample's decode_with2d dict against the golden. Detections are compared as SETS: the final
    ranking sorts scores that can sit 1e-7 apart, and the 2D slot order follows the (tie-broken) bank
    order, so rows may legitimately come out in a different order ('compare sets with
    tolerance, not positions'). Every row must have a partner with box within box_tol, score within
    score_tol and the same label; the 2D<->3D association must be the same relation under that pairing."""
    def G(k):
        return g[prefix + k]

    def T(k):
        v = res[k]
        return np.asarray(v.detach().cpu() if torch.is_tensor(v) else v)

    def rec3(b, s, c, l):
        b = np.asarray(b, np.float64)
        return np.concatenate([b[:, :6], np.sin(b[:, 6:7]), np.cos(b[:, 6:7]), b[:, 7:], np.asarray(s, np.float64)[:, None],
                               np.asarray(c, np.float64)[:, None], np.asarray(l, np.float64)[:, None]], axis=1)

    tol3 = [box_tol] * 11 + [score_tol, score_tol, 0.5]
    p3 = _match_rows(rec3(T("boxes_3d"), T("scores_3d"), T("cls_scores"), T("labels_3d")),
                     rec3(G("boxes_3d"), G("scores_3d"), G("cls_scores"), G("labels_3d")), tol3)
    assert p3 is not None, "3D detections differ (as a set) beyond tolerance"
    # ranking may only differ where scores are tied within tolerance
    assert np.abs(T("scores_3d").astype(np.float64) - G("scores_3d")).max() <= score_tol
    rec2 = lambda b, s, l: np.concatenate([np.asarray(b, np.float64), np.asarray(s, np.float64)[:, None],
                                           np.asarray(l, np.float64)[:, None]], axis=1)
    tol2 = [box_tol * 100] * 4 + [score_tol, 0.5]
    p2 = _match_rows(rec2(T("boxes_2d"), T("scores_2d"), T("labels_2d")),
                     rec2(G("boxes_2d"), G("scores_2d"), G("labels_2d")), tol2)
    assert p2 is not None, "2D detections differ (as a set) beyond tolerance"
    # camidx_2d is bucketed with the reference's re-bound group list (decoder.py:216), so for sample
    # i > 0 of a batch its length need not equal the number of 2D boxes: compared as it stands
    assert np.array_equal(np.sort(T("camidx_2d").astype(np.int64)), np.sort(G("camidx_2d").astype(np.int64)))
    t = res["trans_matrix"]
    t = t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)
    assert tuple(t.shape) == tuple(G("trans_shape").tolist())
    inv3 = np.empty_like(p3); inv3[p3] = np.arange(len(p3))  # got row -> want row
    inv2 = np.empty_like(p2); inv2[p2] = np.arange(len(p2))
    nz = torch.nonzero(t).numpy()
    got_pairs = {(int(inv3[r]), int(inv2[c])) for r, c in nz}
    assert got_pairs == {(int(r), int(c)) for r, c in G("trans_nz")}
    assert np.array_equal(np.asarray(res["query_groups"], np.int64), G("query_groups").astype(np.int64))
    ids = T("instance_ids").astype(np.int64)[p3]
    want_ids = G("instance_ids").astype(np.int64)
    if not np.array_equal(ids, want_ids):  # same tracks up to a relabelling (see rows_match)
        pairs = set(zip(ids.tolist(), want_ids.tolist()))
        assert len(pairs) == len(set(ids.tolist())) == len(set(want_ids.tolist())), "instance ids are not a relabelling"


def _flat(x):
    from simpb_amd.plugin.dense import Segments
    if isinstance(x, Segments):  # an unmaterialised residual_mode="cat" result
        return [x.materialize()]
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, (list, tuple)):
        out = []
        for y in x:
            out += _flat(y)
        return out
    return []


def attach_trace_hooks(head, trace):
    """Forward hooks on the product head at the same module boundaries the golden generator
    hooked on the reference head (tools/golden/gen_golden.py:attach_hooks), so records line up
    by name. The allocation record is taken in index form from `layer.last`."""
    hooks = []
    for i, (op, layer) in enumerate(zip(head.operation_order, head.layers)):
        if layer is None:
            continue
        name = f"L{i:02d}.{op}"

        def hook(mod, inp, out, name=name, op=op):
            if op == "allocation":
                pts, depth, tmask, tshape = out[:4]
                a = mod.last
                trace.add(name + ".ref_pts2d", pts)
                trace.add(name + ".ref_depth2d", depth)
                trace.add(name + ".trans_mask", tmask)
                trace.add(name + ".trans_shape", tshape)
                trace.add(name + ".q2a", a.q2a)
                trace.add(name + ".is_center", a.is_center)
                trace.add(name + ".query_groups", torch.tensor(a.query_groups, dtype=torch.int32))
            else:
                for k, t in enumerate(_flat(out)):
                    trace.add(f"{name}.{k}", t)

        hooks.append(layer.register_forward_hook(hook))
    for nm in ("anchor_encoder", "anchor_encoder2d", "fc_after", "fc_after2d"):
        hooks.append(getattr(head, nm).register_forward_hook(lambda m, i, o, nm=nm: trace.add(nm, o)))
    return hooks


def build_product_head(spec, device="cuda"):
    """The product head for a golden spec: shipped config with the spec's anchor/bank/decoder sizes
    (the same edits tools/golden/gen_golden.py:build_ref_head makes) and procedural weights."""
    from simpb_amd import configs, plugin
    cfg = configs.simpb_plus(anchor=synth.anchors(spec["num_anchor"]))["model"]["head"]
    cfg["instance_bank"]["num_anchor"] = spec["num_anchor"]
    cfg["instance_bank"]["num_temp_instances"] = spec["num_temp"]
    cfg["num_anchor"] = spec["num_anchor"]
    cfg["decoder"] = dict(type="SparseBox3DDecoder", num_output=spec["num_output"])
    head = plugin.build_head(cfg).eval()
    synth.load_procedural(head)
    return head.to(device)


def metas_to(metas, device):
    out = dict(metas)
    for k in ("projection_mat", "image_wh", "timestamp"):
        out[k] = metas[k].to(device)
    out["image_wh_host"] = tuple(int(v) for v in metas["image_wh"][0, 0].tolist())
    return out
