"""CPU: the oracle (oracle/simpb_ref.py) against the golden vectors captured from the reference's
own model files (tools/golden/gen_golden.py). This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import simpb_ref as R
from simpb_amd import synth
from tests.helpers import compare_result, compare_trace, golden_params, load_golden, spec_of


def test_feature_maps_format():
    g = load_golden("ops.npz")
    maps = [torch.from_numpy(synth.randn(f"ops.fmt.l{l}", (2, 6, 8, h, w))) for l, (h, w) in
            enumerate([(4, 6), (2, 3), (1, 2)])]
    col, ss, ssi = R.feature_maps_format(maps)
    assert torch.equal(col, torch.from_numpy(g["fmt.col"]))
    assert np.array_equal(ss.numpy(), g["fmt.spatial_shape"])
    assert np.array_equal(ssi.numpy(), g["fmt.scale_start_index"])


def daf_case(g):
    shapes = [tuple(s) for s in g["daf.shapes"].tolist()]
    fmaps = [torch.from_numpy(synth.randn(f"ops.daf.l{l}", (2, 6, 16, h, w))) for l, (h, w) in enumerate(shapes)]
    col, ss, ssi = R.feature_maps_format(fmaps)
    return col, ss, ssi, torch.from_numpy(g["daf.loc"]), torch.from_numpy(g["daf.weights"])


def test_daf_vs_reference_fallback():
    """Interior points: the CUDA kernel's semantics and the reference's grid_sample fallback
    (blocks.py:149-156) coincide, so the fallback's output pins the bilinear arithmetic."""
    g = load_golden("ops.npz")
    col, ss, ssi, loc, w = daf_case(g)
    out = R.deformable_aggregation(col, ss, ssi, loc, w)
    assert np.abs(out.numpy() - g["daf.out_fallback"]).max() < 1e-5


def test_daf_known_answers():
    """deformable_aggregation_cuda.cu:169-171: samples on or outside the (0,1) border vanish;
    a constant map returns the sum of the valid weights (taps inside the map)."""
    bs, A, P, K, L, G, C = 1, 3, 2, 2, 2, 2, 8
    maps = [torch.ones(bs, K, C, 6, 8), torch.ones(bs, K, C, 3, 4)]
    col, ss, ssi = R.feature_maps_format(maps)
    loc = torch.full((bs, A, P, K, 2), 0.5)
    loc[0, 1, :, :, 0] = 0.0  # x == 0 -> dropped
    loc[0, 2, :, :, 1] = 1.0  # y == 1 -> dropped
    w = torch.rand(bs, A, P, K, L, G)
    out = R.deformable_aggregation(col, ss, ssi, loc, w)
    want = w[0, 0].sum(dim=(0, 1, 2)).repeat_interleave(C // G)
    assert torch.allclose(out[0, 0], want, atol=1e-6)
    assert out[0, 1].abs().max() == 0 and out[0, 2].abs().max() == 0


def test_msda_grouped_vs_reference_loop():
    """[parity unpinned: the sampler is mmcv's] the reference's own per-camera loop
    (group_attn.py:227-235) over the restated sampler."""
    g = load_golden("ops.npz")
    keys = ["sampling_offsets", "attention_weights", "value_proj", "output_proj"]
    shp = {"sampling_offsets": (256, 256), "attention_weights": (128, 256), "value_proj": (256, 256), "output_proj": (256, 256)}
    p = {}
    for k in keys:
        p[f"m.{k}.weight"] = torch.from_numpy(synth.procedural_tensor(f"{k}.weight", shp[k]))
        p[f"m.{k}.bias"] = torch.from_numpy(synth.procedural_tensor(f"{k}.bias", shp[k][:1]))
    shapes = g["msda.shapes"]
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    q = torch.from_numpy(synth.randn("ops.msda.q", (2, 52, 256)))
    qpos = torch.from_numpy(synth.randn("ops.msda.qpos", (2, 52, 256)))
    val = torch.from_numpy(synth.randn("ops.msda.value", (12, nv, 256)))
    enc = dict(value=val, spatial_shapes=torch.from_numpy(shapes).long())
    out = R.qg_msda(p, "m", q, qpos, torch.from_numpy(g["msda.ref"]), [tuple(x) for x in g["msda.groups"].tolist()], enc)
    assert np.abs(out.numpy() - g["msda.out"]).max() < 2e-5


def test_allocation_known_answers():
    g = load_golden("ops.npz")
    metas = synth.frame_metas(1, 0)
    pts, depth, tmask, tshape, trans, cmat, groups = R.allocation(torch.from_numpy(g["alloc.anchor"]), metas)
    assert np.array_equal(tshape.numpy(), g["alloc.trans_shape"])
    assert np.array_equal(np.asarray(groups), g["alloc.query_groups"])
    assert np.array_equal(tmask.numpy(), g["alloc.trans_mask"])
    assert np.allclose(pts.numpy(), g["alloc.ref_pts2d"], atol=1e-6)
    assert np.allclose(depth.numpy(), g["alloc.ref_depth2d"], atol=1e-5)
    assert np.array_equal(torch.where(trans.sum(-1) > 0, trans.argmax(-1), -1).numpy(), g["alloc.q2a"])
    assert np.array_equal(cmat.sum(-1).numpy().astype(np.int32), g["alloc.is_center"])


def run_oracle_stream(g, frames=None, with_trace=True):
    spec = spec_of(g)
    params = golden_params(g)
    head = R.OracleHead(params, g["operation_order"].tolist(), spec["num_anchor"], spec["num_temp"], spec["num_output"])
    out = []
    with torch.no_grad():
        for f in range(spec["frames"] if frames is None else frames):
            head.trace = synth.Trace() if with_trace else None
            fm = R.feature_maps_format(synth.feature_maps_nchw(spec["bs"], f, spec["image_wh"]))
            metas = synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"])
            outs = head.forward(fm, metas)
            res = head.post_process(outs, metas)
            out.append((head.trace, outs, res, head.bank))
            yield f, head.trace, outs, res, head.bank


@pytest.mark.parametrize("name", ["head_small.npz", "head_r50.npz"])
def test_head_stream(name):
    """Whole head, consecutive frames (temporal bank, ego-motion warp, bs=2 padding and a time
    gap in the small case; the shipped R50 704x256 shapes in the other)."""
    g = load_golden(name)
    spec = spec_of(g)
    torch.set_num_threads(8)
    for f, trace, outs, res, bank in run_oracle_stream(g):
        pre = f"f{f}."
        assert [x.shape[1] for x in outs["prediction2d"]] == g[pre + "n2#0"].tolist()
        if f in spec["trace_frames"]:
            compare_trace(trace, g, pre + "trace.")
        assert np.allclose(bank.cached_anchor.numpy(), g[pre + "bank.cached_anchor#0"], atol=2e-4)
        assert np.array_equal(bank.instance_id.numpy(), g[pre + "bank.instance_id#0"])
        assert np.array_equal(outs["instance_id"].numpy(), g[pre + "instance_id#0"])
        for b, r in enumerate(res):
            compare_result(r, g, f"{pre}res{b}.")


def test_c_restatement_of_the_native_kernel():
    """oracle/daf_ref.c (line-by-line C form of the CUDA kernel) against the reference's own PyTorch
    fallback on interior points and against the PyTorch restatement on border/out-of-range points."""
    from oracle import build_c
    g = load_golden("ops.npz")
    col, ss, ssi, loc, w = daf_case(g)
    out = build_c.daf_forward(col.numpy(), ss.numpy(), ssi.numpy(), loc.numpy(), w.numpy())
    assert np.abs(out - g["daf.out_fallback"]).max() < 1e-5
    rs = np.random.RandomState(4)
    loc2 = torch.from_numpy(rs.uniform(-0.2, 1.2, loc.shape).astype(np.float32))
    loc2[0, 0, 0, 0] = torch.tensor([0.0, 0.5])
    loc2[0, 1, 0, 0] = torch.tensor([0.5, 1.0])
    want = R.deformable_aggregation(col, ss.int(), ssi.int(), loc2, w)
    got = build_c.daf_forward(col.numpy(), ss.numpy(), ssi.numpy(), loc2.numpy(), w.numpy())
    assert np.abs(got - want.numpy()).max() < 1e-5
