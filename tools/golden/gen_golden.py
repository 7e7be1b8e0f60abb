"""Generate tests/golden/*.npz by running the reference's own model files (through ref_shim) in
the build container. Usage: python tools/golden/gen_golden.py [--only NAME]

The fixtures hold inputs that are not procedural plus expected outputs; weights and feature maps
are rebuilt on both sides from simpb_amd.synth. Nothing of the reference's source is stored.
"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from simpb_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

SMALL = dict(image_wh=(176, 64), num_anchor=48, num_temp=32, num_output=20, bs=2, frames=3, jump=(1, 2, 5.0), trace_frames=(0, 1, 2))
R50 = dict(image_wh=(704, 256), num_anchor=900, num_temp=600, num_output=300, bs=1, frames=4, jump=None, trace_frames=(0, 3))


def build_ref_head(ns, cfg, spec):
    hcfg = ref_shim.head_cfg_for_eval(cfg, synth.anchors(spec["num_anchor"]))
    hcfg["instance_bank"]["num_anchor"] = spec["num_anchor"]
    hcfg["instance_bank"]["num_temp_instances"] = spec["num_temp"]
    hcfg["num_anchor"] = spec["num_anchor"]
    hcfg["decoder"] = dict(type="SparseBox3DDecoder", num_output=spec["num_output"])
    head = ns.head.SimPBHead(**hcfg)
    head.use_deformable_func = True
    head.eval()
    synth.load_procedural(head)
    return head


def _flat(x):
    if torch.is_tensor(x):
        return [x]
    if isinstance(x, (list, tuple)):
        out = []
        for y in x:
            out += _flat(y)
        return out
    return []


def attach_hooks(head, trace):
    hooks = []
    for i, (op, layer) in enumerate(zip(head.operation_order, head.layers)):
        if layer is None:
            continue
        name = f"L{i:02d}.{op}"

        def hook(mod, inp, out, name=name, op=op):
            if op == "allocation":
                pts, depth, tmask, tshape, tmat, cmat, groups, _ = out
                trace.add(name + ".ref_pts2d", pts)
                trace.add(name + ".ref_depth2d", depth)
                trace.add(name + ".trans_mask", tmask)
                trace.add(name + ".trans_shape", tshape)
                trace.add(name + ".q2a", torch.where(tmat.sum(-1) > 0, tmat.argmax(-1), -1).to(torch.int32))
                trace.add(name + ".is_center", cmat.sum(-1).to(torch.int32))
                trace.add(name + ".query_groups", torch.tensor(groups, dtype=torch.int32))
            else:
                for k, t in enumerate(_flat(out)):
                    trace.add(f"{name}.{k}", t)

        hooks.append(layer.register_forward_hook(hook))
    for nm in ("anchor_encoder", "anchor_encoder2d", "fc_after", "fc_after2d"):
        mod = getattr(head, nm)
        hooks.append(mod.register_forward_hook(lambda m, i, o, nm=nm: trace.add(nm, o)))
    return hooks


def pack_result(res, prefix):
    out = {}
    for k, v in res.items():
        if k == "trans_matrix":
            nz = torch.nonzero(v)
            out[prefix + "trans_nz"] = nz.to(torch.int32).numpy()
            out[prefix + "trans_shape"] = np.asarray(v.shape, np.int32)
        elif k == "query_groups":
            out[prefix + k] = np.asarray(v, np.int32)
        else:
            out[prefix + k] = torch.as_tensor(v).cpu().numpy()
    return out


def run_head(ns, cfg, spec, fname):
    torch.manual_seed(0)
    head = build_ref_head(ns, cfg, spec)
    data = {"spec_" + k: np.asarray(v if v is not None else -1) for k, v in spec.items()}
    sd = head.state_dict()
    data["state_keys"] = np.asarray(list(sd.keys()))
    data["state_shapes"] = np.asarray([",".join(map(str, v.shape)) for v in sd.values()])
    data["operation_order"] = np.asarray(head.operation_order)
    with torch.no_grad():
        for f in range(spec["frames"]):
            trace = synth.Trace()
            hooks = attach_hooks(head, trace)
            fm = ns.feature_maps_format(synth.feature_maps_nchw(spec["bs"], f, spec["image_wh"]))
            metas = synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"])
            outs = head(fm, metas)
            res = head.post_process(outs, metas)
            for h in hooks:
                h.remove()
            pre = f"f{f}."
            if f in spec["trace_frames"]:
                data.update(trace.as_npz_dict(pre + "trace."))
            bank = head.instance_bank
            trace2 = synth.Trace()
            trace2.add("bank.cached_anchor", bank.cached_anchor)
            trace2.add("bank.cached_feature", bank.cached_feature)
            trace2.add("bank.confidence", bank.confidence)
            trace2.add("bank.instance_id", bank.instance_id)
            trace2.add("instance_id", outs["instance_id"])
            trace2.add("n2", torch.tensor([x.shape[1] for x in outs["prediction2d"]]))
            data.update(trace2.as_npz_dict(pre))
            for b, r in enumerate(res):
                data.update(pack_result(r["img_bbox"], f"{pre}res{b}."))
            print(fname, "frame", f, "N2", [x.shape[1] for x in outs["prediction2d"]],
                  "top score", float(res[0]["img_bbox"]["scores_3d"][0]))
    path = os.path.join(OUT, fname)
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(data), "arrays")


def run_ops(ns, cfg):
    """Stand-alone vectors for the two samplers and the format function."""
    data = {}
    DFA = ns.blocks.DeformableFeatureAggregation

    # (1) feature_maps_format (ops/__init__.py:63-92)
    maps = [torch.from_numpy(synth.randn(f"ops.fmt.l{l}", (2, 6, 8, h, w))) for l, (h, w) in
            enumerate([(4, 6), (2, 3), (1, 2)])]
    col, ss, ssi = ns.feature_maps_format(maps)
    data.update({"fmt.col": col.numpy(), "fmt.spatial_shape": ss.numpy(), "fmt.scale_start_index": ssi.numpy()})

    # (2) DAF pinned by the reference's own PyTorch fallback (blocks.py:149-156,215-261) on points
    # strictly inside the image, where the kernel's (0,1) gate and zero-padded grid_sample agree.
    bs, A, P, K, L, G, C = 2, 10, 5, 6, 3, 4, 16
    shapes = [(8, 12), (4, 6), (2, 3)]
    fmaps = [torch.from_numpy(synth.randn(f"ops.daf.l{l}", (bs, K, C, h, w))) for l, (h, w) in enumerate(shapes)]
    # key points (x, y, 1) with per-camera projection diag(s_cam) so that the reference's
    # project_points (blocks.py:198-213) yields u[cam] = s_cam * base, all inside (0,1).
    base = np.random.RandomState(1).uniform(0.002, 0.998, (bs, A, P, 2)).astype(np.float32)
    key_points = torch.from_numpy(np.concatenate([base, np.ones((bs, A, P, 1), np.float32)], -1))
    s_cam = torch.tensor([1.0, 0.9, 0.75, 0.6, 0.45, 0.3])
    proj = torch.zeros(bs, K, 4, 4)
    proj[:, :, 0, 0] = s_cam
    proj[:, :, 1, 1] = s_cam
    proj[:, :, 2, 2] = 1.0
    proj[:, :, 3, 3] = 1.0
    wts = torch.from_numpy(np.random.RandomState(2).uniform(0, 1, (bs, A, K, L, P, G)).astype(np.float32))

    class Dummy:
        num_groups, group_dims, num_pts, embed_dims = G, C // G, P, C

    u = DFA.project_points(key_points, proj, None)  # [bs, cam, A, P, 2]
    feats = DFA.feature_sampling(fmaps, key_points, proj, None)
    fused = DFA.multi_view_level_fusion(Dummy, feats, wts).sum(dim=2)
    col, ss, ssi = ns.feature_maps_format(fmaps)
    loc = u.permute(0, 2, 3, 1, 4).contiguous()  # [bs, A, P, cam, 2]
    w_kernel = wts.permute(0, 1, 4, 2, 3, 5).contiguous()  # [bs, A, P, cam, lvl, G]
    mine = ref_shim.daf_kernel_semantics(col, ss, ssi, loc, w_kernel)
    err = (mine - fused).abs().max().item()
    print("DAF stand-in vs reference fallback (interior points): max abs", err)
    assert err < 1e-5
    data.update({"daf.loc": loc.numpy(), "daf.weights": w_kernel.numpy(), "daf.out_fallback": fused.numpy(),
                 "daf.shapes": np.asarray(shapes, np.int32)})

    # (3) grouped MSDA through the reference's own per-camera loop (group_attn.py:146-256) over the
    # restated mmcv sampler [mmcv-memory -> parity unpinned].
    msda = ns.group_attn.QueryGroupMultiScaleDeformableAttention(
        batch_first=True, num_levels=4, embed_dims=256, num_points=4, residual_mode="cat").eval()
    synth.load_procedural(msda)
    shapes = [(8, 22), (4, 11), (2, 6), (1, 3)]
    nv = sum(h * w for h, w in shapes)
    groups = [(0, 7), (7, 7), (7, 20), (20, 31), (31, 40), (40, 52)]
    nq, bs = 52, 2
    q = torch.from_numpy(synth.randn("ops.msda.q", (bs, nq, 256)))
    qpos = torch.from_numpy(synth.randn("ops.msda.qpos", (bs, nq, 256)))
    val = torch.from_numpy(synth.randn("ops.msda.value", (bs * 6, nv, 256)))
    ref = torch.from_numpy(np.random.RandomState(3).uniform(-0.1, 1.1, (bs, nq, 2)).astype(np.float32))
    ss = torch.tensor(shapes)
    lsi = torch.cat([ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]])
    with torch.no_grad():
        out = msda(query=q, query_pos=qpos, reference_points=ref.unsqueeze(2), query_groups=groups, value=val,
                   key_padding_mask=torch.zeros(bs * 6, nv, dtype=torch.bool), spatial_shapes=ss,
                   level_start_index=lsi)
    data.update({"msda.ref": ref.numpy(), "msda.groups": np.asarray(groups, np.int32),
                 "msda.shapes": np.asarray(shapes, np.int32), "msda.out": out.numpy()})

    # (4) allocation known-answer cases (allocation.py:27-144): one anchor dead ahead of camera 0,
    # one behind every camera's image, one straddling the seam of two cameras.
    alloc = ns.alloc.DynamicQueryAllocation(limit_corners_num=[100] * 6).eval()
    anc = np.zeros((1, 4, 11), np.float32)
    anc[0, :, 3:6] = synth.MEAN_LOG_WLH
    anc[0, :, 7] = 1.0
    anc[0, 0, :3] = (20.0, 0.0, 0.0)
    anc[0, 1, :3] = (0.3, 0.0, 30.0)
    anc[0, 2, :3] = (12.0, 12.0 * np.tan(np.radians(27.5)), 0.0)
    anc[0, 3, :3] = (1.2, 0.2, 0.0)
    metas = synth.frame_metas(1, 0)
    with torch.no_grad():
        pts, depth, tmask, tshape, tmat, cmat, qg, _ = alloc(torch.from_numpy(anc), metas)
    data.update({"alloc.anchor": anc, "alloc.ref_pts2d": pts.numpy(), "alloc.ref_depth2d": depth.numpy(),
                 "alloc.trans_mask": tmask.numpy(), "alloc.trans_shape": tshape.numpy(),
                 "alloc.q2a": torch.where(tmat.sum(-1) > 0, tmat.argmax(-1), -1).numpy().astype(np.int32),
                 "alloc.is_center": cmat.sum(-1).numpy().astype(np.int32), "alloc.query_groups": np.asarray(qg, np.int32)})
    print("alloc known-answer: trans_shape", tshape.tolist(), "groups", qg)

    path = os.path.join(OUT, "ops.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=[None, "ops", "small", "r50"])
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ns = ref_shim.install()
    cfg = ref_shim.load_config()
    if args.only in (None, "ops"):
        run_ops(ns, cfg)
    if args.only in (None, "small"):
        run_head(ns, cfg, SMALL, "head_small.npz")
    if args.only in (None, "r50"):
        run_head(ns, cfg, R50, "head_r50.npz")


if __name__ == "__main__":
    main()
