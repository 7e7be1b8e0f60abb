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
    that index instances are then compared as multisets."""
    names = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    assert names, prefix
    missing = [n for n in names if n not in got.items and not any(n.startswith(s) for s in skip)]
    assert not missing, f"trace records missing: {missing[:5]}"
    bad = []
    for n in names:
        if any(n.startswith(s) for s in skip):
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


def compare_result(res, g, prefix, box_tol=1e-3, score_tol=1e-3):
    """One sample's decode_with2d dict against the golden: SURVEY.md §7 'compare sets with
    tolerance' — rows are matched by position after the reference's own score sort; ties in
    top-k would show up as a row mismatch and are reported with the score gap."""
    def G(k):
        return g[prefix + k]

    for k, tol in (("boxes_3d", box_tol), ("scores_3d", score_tol), ("cls_scores", score_tol),
                   ("boxes_2d", box_tol * 100), ("scores_2d", score_tol)):
        a, b = np.asarray(res[k].detach().cpu() if torch.is_tensor(res[k]) else res[k], np.float64), G(k).astype(np.float64)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if k == "boxes_3d":  # yaw wraps
            d = np.abs(a - b)
            d[:, 6] = np.minimum(d[:, 6], 2 * np.pi - d[:, 6])
            err = d.max()
        else:
            err = np.abs(a - b).max() if b.size else 0.0
        assert err <= tol, (k, err, tol)
    for k in ("labels_3d", "labels_2d", "camidx_2d"):
        a = np.asarray(res[k].detach().cpu() if torch.is_tensor(res[k]) else res[k]).astype(np.int64)
        assert np.array_equal(a, G(k).astype(np.int64)), k
    ids = np.asarray(res["instance_ids"].detach().cpu()).astype(np.int64)
    want_ids = G("instance_ids").astype(np.int64)
    if not np.array_equal(ids, want_ids):  # same tracks up to a relabelling (see rows_match)
        pairs = set(zip(ids.tolist(), want_ids.tolist()))
        assert len(pairs) == len(set(ids.tolist())) == len(set(want_ids.tolist())), "instance ids are not a relabelling"
    t = res["trans_matrix"]
    nz = torch.nonzero(t.detach().cpu()).numpy().astype(np.int64)
    assert tuple(t.shape) == tuple(G("trans_shape").tolist())
    assert np.array_equal(nz, G("trans_nz").astype(np.int64))
    assert np.array_equal(np.asarray(res["query_groups"], np.int64), G("query_groups").astype(np.int64))


def _flat(x):
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
