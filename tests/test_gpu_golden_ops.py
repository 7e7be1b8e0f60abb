"""GPU: the reference-generated operator vectors of tests/golden/ops.npz (captured by tools/golden/gen_golden.py from the
reference's own files) run through the HIP kernels themselves, not only through the CPU oracle: token format (A13),
query allocation (A2, bit-exact integers), grouped multi-scale deformable attention (A6), 3D deformable aggregation (A3);
and the allocation tables of the shipped R50 shapes at frame 0, position by position."""
import numpy as np
import pytest
import torch

from simpb_amd import synth
from simpb_amd.plugin import routes
from tests.helpers import build_product_head, load_golden, metas_to, spec_of

pytestmark = pytest.mark.gpu


def _ops():
    from simpb_amd.plugin import ops
    return ops


def test_token_format_vs_reference_vectors():
    """A13, ops/__init__.py:63-92: the product's feature_maps_format AND the one-pass format_tokens kernel (csrc/format.hip,
    fp32 and fp16 sources) against the reference's own output (ops.npz:fmt.*)."""
    ops = _ops()
    g = load_golden("ops.npz")
    shapes = [(4, 6), (2, 3), (1, 2)]
    maps = [torch.from_numpy(synth.randn(f"ops.fmt.l{l}", (2, 6, 8, h, w))) for l, (h, w) in enumerate(shapes)]
    want = torch.from_numpy(g["fmt.col"])
    col, ss, ssi = ops.feature_maps_format([m.cuda() for m in maps])
    assert torch.equal(col.cpu(), want)
    assert np.array_equal(ss.cpu().numpy(), g["fmt.spatial_shape"]) and np.array_equal(ssi.cpu().numpy(), g["fmt.scale_start_index"])
    nhwc = [m.flatten(0, 1).cuda().contiguous(memory_format=torch.channels_last) for m in maps]
    col2, ss2, ssi2 = ops.format_tokens(nhwc, 2, 6)
    assert torch.equal(col2.cpu(), want)
    assert np.array_equal(ss2.cpu().numpy(), g["fmt.spatial_shape"]) and np.array_equal(ssi2.cpu().numpy(), g["fmt.scale_start_index"])
    # fp16 source (what the fp16 backbone hands over): the kernel's fp16 -> fp32 conversion of the same values
    col3 = ops.format_tokens([m.half() for m in nhwc], 2, 6)[0]
    assert torch.equal(col3.cpu(), want.half().float())


def test_allocation_vs_reference_vectors():
    """A2, allocation.py:27-144 on the reference's known-answer case: index tables bit-exact, reference points to 1e-6."""
    from simpb_amd.plugin.allocation import DynamicQueryAllocation
    g = load_golden("ops.npz")
    layer = DynamicQueryAllocation().eval()
    metas = metas_to(synth.frame_metas(1, 0), "cuda")
    pts, depth, tmask, tshape, _, _, groups, _ = layer(torch.from_numpy(g["alloc.anchor"]).cuda(), metas, dense=False)
    a = layer.last
    assert np.array_equal(a.q2a.cpu().numpy(), g["alloc.q2a"])
    assert np.array_equal(a.is_center.cpu().numpy(), g["alloc.is_center"])
    assert np.array_equal(tmask.cpu().numpy(), g["alloc.trans_mask"])
    assert np.array_equal(tshape.cpu().numpy(), g["alloc.trans_shape"])
    assert np.array_equal(np.asarray(groups), g["alloc.query_groups"])
    assert np.abs(pts.cpu().numpy() - g["alloc.ref_pts2d"]).max() <= 1e-6
    assert np.abs(depth.cpu().numpy() - g["alloc.ref_depth2d"]).max() <= 1e-5
    # the static-capacity form of the same call: same tables in the live slots, pads marked
    layer(torch.from_numpy(g["alloc.anchor"]).cuda(), metas, dense=False, capacity=8)
    s = layer.last
    n2 = g["alloc.q2a"].shape[1]
    assert np.array_equal(s.q2a.cpu().numpy()[:, :n2], g["alloc.q2a"]) and (s.q2a.cpu().numpy()[:, n2:] == -1).all()
    assert np.array_equal(s.is_center.cpu().numpy()[:, :n2], g["alloc.is_center"])
    assert int(s.overflow.item()) == 0 and (s.query_cam.cpu().numpy()[n2:] == -1).all()


@pytest.mark.parametrize("bs,capacity", [(1, 1536), (3, 2048), (2, 600)])
def test_static_allocation_call_equals_the_stepwise_kernels(bs, capacity):
    """simpb_alloc_static (the five allocation launches of a replayed frame as three) against the step-by-step kernels the
    reference vectors pin above: every table bit for bit, batched (max-over-batch group table, allocation.py:91-99) and in
    the overflow case (capacity 600 < N2)."""
    from simpb_amd.plugin import allocation
    layer = allocation.DynamicQueryAllocation().eval()
    metas = metas_to(synth.frame_metas(bs, 0), "cuda")
    g = torch.Generator().manual_seed(bs)
    anchor = torch.from_numpy(synth.anchors(900)).float()[None].repeat(bs, 1, 1)
    anchor[..., :2] += torch.randn(bs, 900, 2, generator=g) * 3.0
    anchor = anchor.cuda()
    outs = []
    for fused in (True, False):
        with routes.override(alloc_static_fused=fused):
            a, pts, depth, _, _ = layer.allocate(anchor, metas, capacity=capacity)
        outs.append([a.q2a, a.is_center, a.a2q, a.query_cam, a.count, a.group_start, a.overflow, pts, depth])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert int(outs[0][6].item()) == (1 if capacity == 600 else 0)


@pytest.mark.parametrize("bs,capacity", [(3, 1536), (8, 1280), (2, 600)])
def test_allocation_of_independent_streams_equals_batches_of_one(bs, capacity):
    """simpb_alloc_ragged (per-sample counts in the native path): a batch of independent streams laid out as
    one flat slot array must hold, stream by stream, exactly the tables the pinned static call gives a batch of one
    (allocation.py:27-144 at bs = 1) -- slots shifted by the streams in front, anchors by b * N3, cameras by b * cams --
    and nothing else; the overflow case (capacity 600 < N2) clips every stream like a batch of one."""
    from simpb_amd.plugin import allocation
    layer = allocation.DynamicQueryAllocation().eval()
    metas = metas_to(synth.frame_metas(bs, 0), "cuda")
    g = torch.Generator().manual_seed(bs)
    anchor = torch.from_numpy(synth.anchors(900)).float()[None].repeat(bs, 1, 1)
    anchor[..., :2] += torch.randn(bs, 900, 2, generator=g) * 3.0
    yaw = torch.rand(bs, generator=g) * 6.28   # streams look in different directions: different cameras fill up
    xy = anchor[..., :2].clone()
    anchor[..., 0] = xy[..., 0] * yaw.cos()[:, None] - xy[..., 1] * yaw.sin()[:, None]
    anchor[..., 1] = xy[..., 0] * yaw.sin()[:, None] + xy[..., 1] * yaw.cos()[:, None]
    anchor = anchor.cuda()
    r, pts, depth = layer.allocate_independent(anchor, metas, capacity)
    assert r.streams == bs and tuple(r.q2a.shape) == (1, bs * capacity) and tuple(pts.shape) == (1, bs * capacity, 2)
    gs = r.group_start.cpu().numpy()
    q2a, ctr, cam = r.q2a[0].cpu().numpy(), r.is_center[0].cpu().numpy(), r.query_cam.cpu().numpy()
    a2q = r.a2q.cpu().numpy()
    pts, depth = pts[0].cpu().numpy(), depth[0, :, 0].cpu().numpy()
    assert gs[0] == 0 and (np.diff(gs) >= 0).all()
    any_over = 0
    for b in range(bs):
        one = {k: (v[b:b + 1] if torch.is_tensor(v) else v) for k, v in metas.items() if k != "img_metas"}
        s, spts, sdepth, _, _ = layer.allocate(anchor[b:b + 1], one, capacity=capacity)
        lo, hi = int(gs[b * 6]), int(gs[(b + 1) * 6])
        n = int(s.group_start[6].item())
        any_over |= int(s.overflow.item())
        assert hi - lo == n, (b, lo, hi, n)
        assert np.array_equal(gs[b * 6: b * 6 + 7] - lo, s.group_start.cpu().numpy())
        sq = s.q2a[0, :n].cpu().numpy()
        assert np.array_equal(q2a[lo:hi], np.where(sq >= 0, sq + b * 900, -1))
        assert np.array_equal(ctr[lo:hi], s.is_center[0, :n].cpu().numpy())
        assert np.array_equal(cam[lo:hi], s.query_cam[:n].cpu().numpy() + b * 6)
        assert np.array_equal(pts[lo:hi], spts[0, :n].cpu().numpy()) and np.array_equal(depth[lo:hi], sdepth[0, :n, 0].cpu().numpy())
        sa = s.a2q[0].cpu().numpy()
        assert np.array_equal(a2q[b], np.where(sa >= 0, sa + lo, -1))
    live = int(gs[-1])
    assert (q2a[live:] == -1).all() and (cam[live:] == -1).all() and (ctr[live:] == 0).all()
    assert int(r.overflow.item()) == any_over == (1 if capacity == 600 else 0)


@pytest.mark.parametrize("route", ["linear", "grouped"])
def test_msda_module_vs_reference_vectors(route):
    """A6: QueryGroupMultiScaleDeformableAttention against the reference's own per-camera loop output (ops.npz:msda.out),
    through BOTH routes of the product module, asserting which one ran:
    * linear (what a frame runs, routes.msda_linear; taken when the device-side camera table `query_cam` is passed, as
      SimPBHead passes it): offsets | logits product -> simpb_msda_linear_forward on the RAW tokens -> folded
      W_out . W_value product (csrc/msda_lin.hip, dense.fold_msda_linear);
    * grouped (the drop-in operator's route): value_proj + msda_prep + grouped sampler kernel + output_proj."""
    from simpb_amd.plugin import ops as P
    from simpb_amd.plugin.group_attn import QueryGroupMultiScaleDeformableAttention
    g = load_golden("ops.npz")
    m = QueryGroupMultiScaleDeformableAttention(batch_first=True, embed_dims=256, num_heads=8, num_levels=4, num_points=4,
                                                residual_mode="cat").eval()  # the shipped config (:145-147)
    with torch.no_grad():
        for k in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"):
            lin = getattr(m, k)
            lin.weight.copy_(torch.from_numpy(synth.procedural_tensor(f"{k}.weight", tuple(lin.weight.shape))))
            lin.bias.copy_(torch.from_numpy(synth.procedural_tensor(f"{k}.bias", tuple(lin.bias.shape))))
    m = m.cuda()
    shapes = g["msda.shapes"]
    nv = int((shapes[:, 0] * shapes[:, 1]).sum())
    q = torch.from_numpy(synth.randn("ops.msda.q", (2, 52, 256))).cuda()
    qpos = torch.from_numpy(synth.randn("ops.msda.qpos", (2, 52, 256))).cuda()
    val = torch.from_numpy(synth.randn("ops.msda.value", (12, nv, 256))).cuda()
    ss = torch.from_numpy(shapes).long().cuda()
    lsi = torch.cat([ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]])
    groups = [tuple(x) for x in g["msda.groups"].tolist()]
    ref = torch.from_numpy(g["msda.ref"]).cuda()
    calls = {"linear": [], "grouped": 0}
    orig_lin, orig_grp = P.msda_linear, P.ms_deform_attn_grouped

    def spy_lin(tokens, *a, **kw):
        calls["linear"].append(tokens.dtype)
        return orig_lin(tokens, *a, **kw)

    def spy_grp(*a, **kw):
        calls["grouped"] += 1
        return orig_grp(*a, **kw)
    import simpb_amd.plugin.group_attn as GA
    P.msda_linear, GA.ms_deform_attn_grouped = spy_lin, spy_grp
    extra = dict(query_cam=P.query_cam_from_groups(groups, 52, "cuda")) if route == "linear" else {}
    flat = lambda o: o.materialize() if hasattr(o, "materialize") else o   # noqa: E731
    try:
        with torch.no_grad():
            out = flat(m(query=q, query_pos=qpos, value=val, reference_points=ref.unsqueeze(2), spatial_shapes=ss,
                         level_start_index=lsi, query_groups=groups, key_padding_mask=None, **extra))
            tol = 2e-5 * max(1.0, float(np.abs(g["msda.out"]).max()))
            assert np.abs(out.cpu().numpy() - g["msda.out"]).max() < tol
            if route == "linear":
                assert calls["linear"] == [torch.float32] and calls["grouped"] == 0
                # the f16 token copy the FPN leaves (value_f16): TOK = _Float16 on tokens that are f16 numbers gives what
                # TOK = float gives on the same numbers widened -- and that is the module's answer on those tokens by the
                # grouped route too (value_proj over every token, the reference's order of operations)
                v16 = val.half()
                wide = flat(m(query=q, query_pos=qpos, value=v16.float(), reference_points=ref.unsqueeze(2), spatial_shapes=ss,
                              level_start_index=lsi, query_groups=groups, **extra))
                half = flat(m(query=q, query_pos=qpos, value=v16.float(), value_f16=v16, reference_points=ref.unsqueeze(2),
                              spatial_shapes=ss, level_start_index=lsi, query_groups=groups, **extra))
                assert calls["linear"] == [torch.float32, torch.float32, torch.float16] and calls["grouped"] == 0
                assert float((half - wide).abs().max()) <= 1e-6 * max(1.0, float(wide.abs().max()))
                grouped = flat(m(query=q, query_pos=qpos, value=v16.float(), reference_points=ref.unsqueeze(2), spatial_shapes=ss,
                                 level_start_index=lsi, query_groups=groups))
                assert calls["grouped"] == 1
                assert float((half - grouped).abs().max()) <= tol
            else:
                assert calls["linear"] == [] and calls["grouped"] == 1
    finally:
        P.msda_linear, GA.ms_deform_attn_grouped = orig_lin, orig_grp


def test_daf_kernel_vs_reference_fallback_vectors():
    """A3 on the reference's interior-point case: the HIP kernel against the output of the reference's own PyTorch
    fallback (blocks.py:149-156,215-261), where the kernel's (0, 1) gate and zero padding agree."""
    ops = _ops()
    g = load_golden("ops.npz")
    shapes = [tuple(s) for s in g["daf.shapes"].tolist()]
    fmaps = [torch.from_numpy(synth.randn(f"ops.daf.l{l}", (2, 6, 16, h, w))).cuda() for l, (h, w) in enumerate(shapes)]
    col, ss, ssi = ops.feature_maps_format(fmaps)
    out = ops.deformable_aggregation_function(col, ss, ssi, torch.from_numpy(g["daf.loc"]).cuda(),
                                              torch.from_numpy(g["daf.weights"]).cuda())
    assert np.abs(out.cpu().numpy() - g["daf.out_fallback"]).max() < 1e-5


def test_allocation_tables_r50_frame0_position_by_position():
    """The first allocation layer of the shipped R50 704x256 stream at frame 0 (900 learned anchors, 6 cameras): every
    integer table equals the reference's element by element (no set-wise comparison), reference points to 1e-5, in the
    exact-size form and in the static-capacity form the runners use."""
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    head = build_product_head(spec)
    metas = metas_to(synth.frame_metas(1, 0, spec["image_wh"]), "cuda")
    anchor = head.instance_bank.anchor.detach()[None]
    layer = head.layers[head.operation_order.index("allocation")]
    G = lambda k: g[f"f0.trace.L00.allocation.{k}#0"]  # noqa: E731
    with torch.no_grad():
        pts, depth, tmask, tshape, _, _, groups, _ = layer(anchor, metas, dense=False)
        a = layer.last
        assert np.array_equal(a.q2a.cpu().numpy(), G("q2a")) and np.array_equal(a.is_center.cpu().numpy(), G("is_center"))
        assert np.array_equal(tmask.cpu().numpy().astype(np.uint8), G("trans_mask"))
        assert np.array_equal(tshape.cpu().numpy(), G("trans_shape")) and np.array_equal(np.asarray(groups), G("query_groups"))
        assert np.abs(pts.cpu().numpy() - G("ref_pts2d")).max() <= 1e-5
        n2 = G("q2a").shape[1]
        layer(anchor, metas, dense=False, capacity=1536)
        s = layer.last
        assert np.array_equal(s.q2a.cpu().numpy()[:, :n2], G("q2a")) and (s.q2a.cpu().numpy()[:, n2:] == -1).all()
        assert np.array_equal(s.is_center.cpu().numpy()[:, :n2], G("is_center"))
        assert np.array_equal(s.group_start.cpu().numpy()[:-1], G("query_groups")[:, 0])
        assert np.array_equal(s.count.cpu().numpy().astype(np.int64), G("trans_shape")) and int(s.overflow.item()) == 0
