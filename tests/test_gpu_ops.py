"""GPU: the two hand-written samplers, called through the C-ABI, against the oracle on the same
seeded inputs and against the golden vectors. Tolerances are stated per test; both sides are fp32
and differ only in summation order."""
import numpy as np
import pytest
import torch

from simpb_amd import synth
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


def _ops():
    from simpb_amd.plugin import ops
    return ops


def _oracle():
    from oracle import simpb_ref
    return simpb_ref


def test_library_loaded():
    from simpb_amd import _lib
    assert _lib.lib().simpb_abi_version() == 7


def test_daf_golden_fallback_case():
    from tests.test_oracle_golden import daf_case
    g = load_golden("ops.npz")
    col, ss, ssi, loc, w = daf_case(g)
    out = _ops().deformable_aggregation_function(col.cuda(), ss.cuda(), ssi.cuda(), loc.cuda(), w.cuda())
    assert np.abs(out.cpu().numpy() - g["daf.out_fallback"]).max() < 1e-5


@pytest.mark.parametrize("shape", [
    dict(bs=2, A=37, P=13, K=6, L=4, G=8, C=256, maps=[(16, 44), (8, 22), (4, 11), (2, 6)]),   # fast path, shipped C/G
    dict(bs=1, A=5, P=3, K=2, L=2, G=4, C=64, maps=[(5, 7), (3, 2)]),                       # fast path, partial wave
    dict(bs=2, A=9, P=4, K=3, L=5, G=3, C=30, maps=[(6, 5), (3, 3), (2, 2), (1, 1), (1, 3)]),  # generic path, 5 levels
    dict(bs=1, A=4, P=25, K=6, L=1, G=8, C=256, maps=[(8, 8)]),                              # P*K > 128 -> generic
])
def test_daf_random_vs_oracle(shape):
    """Locations drawn from [-0.2, 1.2] so the (0,1) gate, the border taps and the zero padding
    are all exercised; plus exact 0 and 1."""
    R = _oracle()
    s = shape
    rs = np.random.RandomState(7)
    maps = [torch.from_numpy(rs.standard_normal((s["bs"], s["K"], s["C"], h, w)).astype(np.float32)) for h, w in s["maps"]]
    col, ss, ssi = R.feature_maps_format(maps)
    loc = torch.from_numpy(rs.uniform(-0.2, 1.2, (s["bs"], s["A"], s["P"], s["K"], 2)).astype(np.float32))
    loc[0, 0, 0, 0, 0] = 0.0
    loc[0, 1, 0, 0, 1] = 1.0
    loc[0, 2, 0, 0] = torch.tensor([1e-6, 1 - 1e-6])
    w = torch.from_numpy(rs.uniform(0, 1, (s["bs"], s["A"], s["P"], s["K"], s["L"], s["G"])).astype(np.float32))
    want = R.deformable_aggregation(col, ss.int(), ssi.int(), loc, w)
    got = _ops().deformable_aggregation_function(col.cuda(), ss.cuda(), ssi.cuda(), loc.cuda(), w.cuda()).cpu()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0)


def test_daf_full_size_properties():
    """BASELINE.json config #2 shapes (89 760 tokens x 256): linearity in the weights and the
    constant-map identity (out = sum of valid weights), which need no oracle run at this size,
    then the oracle itself on the same input."""
    R = _oracle()
    ops = _ops()
    fm = R.feature_maps_format(synth.feature_maps_nchw(1, 0))
    col, ss, ssi = fm[0].cuda(), fm[1].cuda(), fm[2].cuda()
    rs = np.random.RandomState(3)
    loc = torch.from_numpy(rs.uniform(-0.5, 1.5, (1, 900, 13, 6, 2)).astype(np.float32)).cuda()
    w1 = torch.from_numpy(rs.uniform(0, 1, (1, 900, 13, 6, 4, 8)).astype(np.float32)).cuda()
    w2 = torch.from_numpy(rs.uniform(0, 1, (1, 900, 13, 6, 4, 8)).astype(np.float32)).cuda()
    o1 = ops.deformable_aggregation_function(col, ss, ssi, loc, w1)
    o2 = ops.deformable_aggregation_function(col, ss, ssi, loc, w2)
    o12 = ops.deformable_aggregation_function(col, ss, ssi, loc, w1 + 2 * w2)
    assert float((o12 - (o1 + 2 * o2)).abs().max()) < 2e-3
    assert torch.equal(o1, ops.deformable_aggregation_function(col, ss, ssi, loc, w1))  # deterministic
    ones = torch.ones_like(col)
    keep = ((loc > 0) & (loc < 1)).all(-1)
    oc = ops.deformable_aggregation_function(ones, ss, ssi, loc, w1)
    upper = (w1 * keep[..., None, None]).sum(dim=(2, 3, 4)).repeat_interleave(32, dim=-1)
    assert bool((oc <= upper + 1e-3).all())  # border taps only remove mass
    want = R.deformable_aggregation(fm[0], fm[1].int(), fm[2].int(), loc.cpu(), w1.cpu())
    assert float((o1.cpu() - want).abs().max()) <= 1e-4 * max(float(want.abs().max()), 1.0)
    from oracle import build_c  # the line-by-line C form of the reference kernel, same input
    want_c = build_c.daf_forward(fm[0].numpy(), fm[1].numpy(), fm[2].numpy(), loc.cpu().numpy(), w1.cpu().numpy())
    assert float(np.abs(o1.cpu().numpy() - want_c).max()) <= 1e-4 * max(float(np.abs(want_c).max()), 1.0)


def _msda_inputs(bs, nq, heads, ch, shapes, pts, ncam, seed):
    rs = np.random.RandomState(seed)
    nv = sum(h * w for h, w in shapes)
    value = torch.from_numpy(rs.standard_normal((bs, ncam, nv, heads, ch)).astype(np.float32))
    loc = torch.from_numpy(rs.uniform(-0.3, 1.3, (bs, nq, heads, len(shapes), pts, 2)).astype(np.float32))
    aw = torch.from_numpy(rs.uniform(0, 1, (bs, nq, heads, len(shapes), pts)).astype(np.float32))
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat([ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]])
    bounds = sorted(rs.choice(np.arange(nq + 1), ncam - 1).tolist())
    groups = list(zip([0] + bounds, bounds + [nq]))
    return value, ss, lsi, loc, aw, groups


@pytest.mark.parametrize("cfg", [
    dict(bs=2, nq=53, heads=8, ch=32, shapes=[(8, 22), (4, 11), (2, 6), (1, 3)], pts=4, ncam=6),  # shipped layout
    dict(bs=1, nq=17, heads=4, ch=8, shapes=[(5, 4), (2, 3)], pts=3, ncam=3),                    # generic points
    dict(bs=1, nq=9, heads=16, ch=32, shapes=[(6, 6), (3, 3), (2, 2), (1, 1), (1, 2)], pts=8, ncam=2),  # 512 channels
])
def test_msda_grouped_vs_oracle(cfg):
    R = _oracle()
    value, ss, lsi, loc, aw, groups = _msda_inputs(seed=11, **cfg)
    outs = []
    for i, (s, e) in enumerate(groups):
        if e > s:
            outs.append(R.ms_deform_attn(value[:, i].contiguous(), ss, loc[:, s:e].contiguous(), aw[:, s:e].contiguous()))
    want = torch.cat(outs, dim=1)
    ops = _ops()
    qcam = ops.query_cam_from_groups(groups, cfg["nq"], "cuda")
    got = ops.ms_deform_attn_grouped(value.cuda(), ss.cuda(), lsi.cuda(), loc.cuda(), aw.cuda(), qcam).cpu()
    assert float((got - want).abs().max()) <= 2e-5 * max(float(want.abs().max()), 1.0)


def _msda_linear_case(seed, bs, nq, ncam, shapes, live, edge=True):
    """Inputs of QueryGroupMultiScaleDeformableAttention.forward between its projections (group_attn.py:176-243): raw camera
    tokens that ARE f16 numbers (what the fp16 FPN leaves), the [offsets | logits] row of every query, 2-d reference points
    placed ON and just beyond every edge of the normalised image, offsets of up to +-3 pixels of each level -- so a large
    share of the bilinear taps falls outside the map and the bias term of the linearity rewrite
    (b_h . sum of valid tap weights, csrc/msda_lin.hip) differs visibly from a plain b_h."""
    rs = np.random.RandomState(seed)
    nv = sum(h * w for h, w in shapes)
    tokens = torch.from_numpy(rs.standard_normal((bs, ncam, nv, 256)).astype(np.float32)).half().float()
    raw = torch.from_numpy(np.concatenate([rs.uniform(-3, 3, (bs, nq, 256)), rs.standard_normal((bs, nq, 128)) * 2], -1).astype(np.float32))
    ref = torch.from_numpy(rs.uniform(0.0, 1.0, (bs, nq, 2)).astype(np.float32))
    if edge:   # every map edge and corner, exactly on it / one texel inside / beyond it
        marks = [0.0, 1.0, 1e-3, 1 - 1e-3, -0.02, 1.02, 0.5]
        k = 0
        for mx in marks:
            for my in marks:
                if k < nq:
                    ref[:, k, 0], ref[:, k, 1] = mx, my
                    k += 1
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat([ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]])
    bounds = sorted(rs.choice(np.arange(live + 1), ncam - 1).tolist())
    groups = list(zip([0] + bounds, bounds + [live]))   # camera groups over the live slots; [live, nq) are capacity slots
    return tokens, raw, ref, ss, lsi, groups


@pytest.mark.parametrize("tok", ["f16", "f32"])
@pytest.mark.parametrize("use_m_live", [True, False], ids=["m_live", "no_m_live"])
def test_msda_linear_route_vs_oracle(tok, use_m_live):
    """The sampler a frame actually runs (simpb_msda_linear_forward, csrc/msda_lin.hip, TOK = _Float16 and float) together
    with the folded [256, 2176] product behind it (dense.fold_msda_linear) against the oracle's
    value_proj -> per-camera sampler -> output_proj (oracle/simpb_ref.py: qg_msda's arithmetic, group_attn.py:176-243) on
    the same inputs: reference points on / beyond every map edge, capacity slots (query_cam = -1), a device-side live
    count. Operator tolerance 2e-5 . max|ref|."""
    import torch.nn as nn
    R = _oracle()
    ops = _ops()
    from simpb_amd.plugin import dense
    # (the device-side live count is a row count of the flat [bs * slots] operand, which the head only passes for a batch
    # of one or for the flat slot array of independent streams: SimPBHead._m_live)
    bs, nq, ncam, live = (1 if use_m_live else 2), 96, 6, 83
    shapes = [(8, 22), (4, 11), (2, 6), (1, 3)]
    tokens, raw, ref, ss, lsi, groups = _msda_linear_case(21, bs, nq, ncam, shapes, live)
    torch.manual_seed(5)
    vp, op = nn.Linear(256, 256), nn.Linear(256, 256)
    with torch.no_grad():
        vp.bias.normal_(0, 0.5)      # a bias of the size of the projected values: the wsum column must be right
        op.bias.normal_(0, 0.1)
    # ---- oracle: project every camera token, sample head slices camera by camera, project out
    with torch.no_grad():
        value = vp(tokens).view(bs, ncam, -1, 8, 32)
        off = raw[..., :256].view(bs, nq, 8, 4, 4, 2)
        aw = raw[..., 256:].view(bs, nq, 8, 16).softmax(-1).view(bs, nq, 8, 4, 4)
        norm = torch.stack([ss[:, 1], ss[:, 0]], -1).float()
        loc = ref[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]
        want = torch.zeros(bs, nq, 256)
        for i, (s, e) in enumerate(groups):
            if e > s:
                want[:, s:e] = op(R.ms_deform_attn(value[:, i].contiguous(), ss, loc[:, s:e].contiguous(), aw[:, s:e].contiguous()))
    # the bias term is non-trivial on this input: with a plain b_h instead of b_h . wsum the answer moves by far more than the tolerance
    with torch.no_grad():
        ones = R.ms_deform_attn(torch.ones(bs, sum(h * w for h, w in shapes), 8, 1), ss, loc[:, :live].contiguous(), aw[:, :live].contiguous())
    assert float((1 - ones).abs().max()) > 0.5 and float((ones < 0.999).float().mean()) > 0.2
    # ---- product: sampler on the raw tokens + folded product
    qcam = torch.full((nq,), -1, dtype=torch.int32)    # the static slot array's table: capacity slots carry camera -1
    for i, (s, e) in enumerate(groups):
        qcam[s:e] = i
    qcam = qcam.cuda()
    assert int((qcam < 0).sum()) == nq - live
    m_live = torch.tensor([live], dtype=torch.int32, device="cuda") if use_m_live else None
    t = tokens.cuda().half() if tok == "f16" else tokens.cuda()
    agg = ops.msda_linear(t.contiguous(), ss.cuda(), lsi.cuda(), raw.cuda(), ref.cuda(), qcam, m_live)
    assert tuple(agg.shape) == (bs, nq, ops.MSDA_LINEAR_WIDTH)
    if not use_m_live:
        assert float(agg[:, live:].abs().max()) == 0.0   # capacity rows hold zeros when the product will read them
    wsum = agg[:, :live, 2048:2056].cpu()
    assert float((wsum - ones).abs().max()) < 1e-5   # [bs, live, 8 heads]: the valid tap weight of every head
    assert float(agg[:, :live, 2056:].abs().max()) == 0.0
    vp, op = vp.cuda(), op.cuda()
    wf, bf = dense.fold_msda_linear(vp, op, 8, ops.MSDA_LINEAR_WIDTH)
    got = dense.linear(agg, wf, bf, m_live=m_live)[:, :live].cpu()
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want[:, :live]).abs().max()) <= 2e-5 * scale, float((got - want[:, :live]).abs().max())


def test_msda_linear_f16_tokens_equal_widened_f32_tokens():
    """TOK = _Float16 and TOK = float read the same numbers and sum them in the same order: equal output rows."""
    ops = _ops()
    bs, nq, ncam, live = 1, 64, 6, 64
    shapes = [(16, 44), (8, 22), (4, 11), (2, 6)]
    tokens, raw, ref, ss, lsi, groups = _msda_linear_case(22, bs, nq, ncam, shapes, live)
    qcam = ops.query_cam_from_groups(groups, nq, "cuda")
    a16 = ops.msda_linear(tokens.cuda().half().contiguous(), ss.cuda(), lsi.cuda(), raw.cuda(), ref.cuda(), qcam)
    a32 = ops.msda_linear(tokens.cuda().contiguous(), ss.cuda(), lsi.cuda(), raw.cuda(), ref.cuda(), qcam)
    assert float((a16 - a32).abs().max()) <= 1e-6 * max(1.0, float(a32.abs().max()))


def test_ops_reject_cpu_tensors():
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.deformable_aggregation_function(torch.zeros(1, 4, 8), torch.ones(1, 1, 2, dtype=torch.int32) * 2,
                                            torch.zeros(1, 1, dtype=torch.int32), torch.zeros(1, 1, 1, 1, 2),
                                            torch.zeros(1, 1, 1, 1, 1, 2))


@pytest.mark.parametrize("m,n,k,relu,bias", [(89760, 256, 256, False, True), (1000, 256, 256, True, True),
                                             (77, 96, 64, False, False), (130, 600, 512, True, True), (1, 256, 32, False, True)])
def test_linear_f32_vs_float64(m, n, k, relu, bias):
    """fp32 MFMA GEMM against a float64 reference: |err| <= 2e-6 * sum_k |x||w| (fp32 chain bound)."""
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((m, k)).astype(np.float32)).cuda()
    w = torch.from_numpy(rs.standard_normal((n, k)).astype(np.float32)).cuda()
    b = torch.from_numpy(rs.standard_normal(n).astype(np.float32)).cuda() if bias else None
    got = _ops().linear_f32(x, w, b, relu=relu)
    want = x.double() @ w.double().t() + (b.double() if bias else 0)
    if relu:
        want = want.clamp(min=0)
    bound = 2e-6 * (x.abs().double() @ w.abs().double().t()) + 1e-6
    assert bool(((got.double() - want).abs() <= bound).all())


def test_mlp_chain_vs_torch_modules():
    """The fused linear_relu_ln chains against the same nn.Sequential evaluated by PyTorch (fp32,
    the tolerance covers summation order only)."""
    import torch.nn as nn
    from simpb_amd.plugin import fused
    from simpb_amd.plugin.detection2d import SparseBox2DEncoder, SparseBox2DRefinementModule
    from simpb_amd.plugin.detection3d import SparseBox3DEncoder, SparseBox3DRefinementModule
    torch.manual_seed(0)
    enc = SparseBox3DEncoder(embed_dims=[128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4)
    synth.load_procedural(enc, seed=3)
    box = torch.randn(2, 77, 11)
    want = torch.cat([enc.pos_fc(box[..., 0:3]), enc.size_fc(box[..., 3:6]), enc.yaw_fc(box[..., 6:8]), enc.vel_fc(box[..., 8:11])], -1)
    got = enc.cuda()(box.cuda()).cpu()
    assert got.shape == want.shape and float((got - want).abs().max()) < 2e-5

    enc2 = SparseBox2DEncoder(embed_dims=256, with_sin_embed=True, in_loops=1, out_loops=2)
    synth.load_procedural(enc2, seed=4)
    pts = torch.rand(1, 203, 2) * 1.2 - 0.1
    want = enc2(pts)
    got = enc2.cuda()(pts.cuda()).cpu()
    assert float((got - want).abs().max()) < 5e-5

    ref3 = SparseBox3DRefinementModule(embed_dims=256, num_cls=10, refine_yaw=True, with_quality_estimation=True)
    synth.load_procedural(ref3, seed=5)
    f, e, a, dt = torch.randn(2, 45, 256), torch.randn(2, 45, 256), torch.randn(2, 45, 11), torch.tensor([0.5, 0.4])
    w_out, w_cls, w_q = ref3(f, a, e, dt, True)
    g_out, g_cls, g_q = ref3.cuda()(f.cuda(), a.cuda(), e.cuda(), dt.cuda(), True)
    for g_, w_ in ((g_out, w_out), (g_cls, w_cls), (g_q, w_q)):
        assert float((g_.cpu() - w_).abs().max()) < 5e-5

    ref2 = SparseBox2DRefinementModule(embed_dims=256, num_cls=10, with_alpha_branch=True)
    synth.load_procedural(ref2, seed=6)
    a2 = torch.rand(2, 45, 2)
    w_box, w_cls, _, w_al = ref2(f, a2, e)
    g_box, g_cls, _, g_al = ref2.cuda()(f.cuda(), a2.cuda(), e.cuda())
    for g_, w_ in ((g_box, w_box), (g_cls, w_cls), (g_al, w_al)):
        assert float((g_.cpu() - w_).abs().max()) < 5e-5


def test_dfa_producers_vs_oracle():
    """dfa_points / dfa_weights against the oracle's key_points -> project_points and dfa_weights
    (same parameters), in the aggregation kernel's layouts."""
    R = _oracle()
    from simpb_amd import configs, plugin
    cfg = configs.simpb_plus(anchor=synth.anchors(900))["model"]["head"]["deformable_model"]
    dfa = plugin.build_from_cfg(cfg, plugin.ATTENTION).eval()
    synth.load_procedural(dfa, seed=2)
    p = {"m." + k: v for k, v in dfa.state_dict().items()}
    bs, A = 2, 61
    feat = torch.from_numpy(synth.randn("dfa.feat", (bs, A, 256)))
    emb = torch.from_numpy(synth.randn("dfa.emb", (bs, A, 256)))
    anchor = torch.from_numpy(synth.anchors(A, seed=4))[None].repeat(bs, 1, 1)
    metas = synth.frame_metas(bs, 0)
    kp = R.key_points(p, "m.kps_generator", anchor, feat)
    want_loc = R.project_points(kp, metas["projection_mat"], metas["image_wh"]).permute(0, 2, 3, 1, 4)
    want_w = R.dfa_weights(p, "m", feat, emb, metas["projection_mat"]).permute(0, 1, 4, 2, 3, 5)
    got = {}
    from simpb_amd.plugin import blocks
    orig = blocks.DAF
    blocks.DAF = lambda f, ss, ssi, loc, w: got.update(loc=loc, w=w) or torch.zeros(bs, A, 256, device="cuda")
    from simpb_amd.plugin import routes
    try:
        dfa.cuda()
        m = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in metas.items()}
        with torch.no_grad(), routes.override(fused_dfa=False):   # the three-launch route, where both operands exist in memory
            dfa(feat.cuda(), anchor.cuda(), emb.cuda(), [None, None, None], m)
    finally:
        blocks.DAF = orig
    assert float((got["loc"].cpu() - want_loc).abs().max()) <= 1e-4 * max(1.0, float(want_loc.abs().max()) * 1e-2)
    assert float((got["w"].cpu() - want_w).abs().max()) <= 1e-6


@pytest.mark.parametrize("f16", [False, True], ids=["f32_tokens", "f16_tokens"])
def test_dfa_fused_launch_equals_three_launches(f16):
    """The shipped one-launch form of DeformableFeatureAggregation (csrc/deform_agg_fused.hip: key points, projection,
    weight softmax and aggregation in one kernel, optionally on the f16 copy of the tokens) against the three-launch form
    through the drop-in operator: same sampling locations and weights (they come back through the optional outputs), same
    module output; and the f16 token copy gives the same bits as the widened rows."""
    from simpb_amd import configs, plugin
    from simpb_amd.plugin import blocks, ops, routes
    cfg = configs.simpb_plus(anchor=synth.anchors(900))["model"]["head"]["deformable_model"]
    dfa = plugin.build_from_cfg(cfg, plugin.ATTENTION).eval().cuda()
    synth.load_procedural(dfa, seed=2)
    bs, A, wh = 2, 77, (352, 128)
    fm = ops.feature_maps_format([x.cuda() for x in synth.feature_maps_nchw(bs, 0, wh)])
    if f16:
        fm[0] = fm[0].half().float()          # tokens that ARE f16 numbers, as the fp16 backbone leaves them
        fm[0].simpb_f16 = fm[0].half()
    fm = [fm[0], fm[1].int().contiguous(), fm[2].int().contiguous()]
    feat = torch.from_numpy(synth.randn("dfa.feat", (bs, A, 256))).cuda()
    emb = torch.from_numpy(synth.randn("dfa.emb", (bs, A, 256))).cuda()
    anchor = torch.from_numpy(synth.anchors(A, seed=4))[None].repeat(bs, 1, 1).cuda()
    anchor[..., :2] *= 0.5   # more key points inside the images
    m = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth.frame_metas(bs, 0, wh).items()}
    seen = {}
    orig, orig_fused = blocks.DAF, blocks.dfa_fused
    blocks.DAF = lambda f, ss, ssi, loc, w: seen.update(loc=loc, w=w) or orig(f, ss, ssi, loc, w)

    def spy(*args, **kw):
        out, loc, w = orig_fused(*args, want_operands=True, **kw)
        seen.update(floc=loc, fw=w, fdtype=args[0].dtype)
        return out
    blocks.dfa_fused = spy
    try:
        with torch.no_grad():
            with routes.override(fused_dfa=False):
                want = dfa(feat, anchor, emb, fm, m)
            with routes.override(dfa_f16_tokens=f16):   # (shipped: the fp32 rows; the f16 copy is the measured alternative)
                got = dfa(feat, anchor, emb, fm, m)
    finally:
        blocks.DAF, blocks.dfa_fused = orig, orig_fused
    assert seen["fdtype"] == (torch.float16 if f16 else torch.float32)
    valid = ((seen["loc"] > 0) & (seen["loc"] < 1)).all(-1)
    assert int(valid.sum()) > 200 and torch.equal(valid, ((seen["floc"] > 0) & (seen["floc"] < 1)).all(-1))
    assert float((seen["floc"] - seen["loc"]).abs().max()) <= 1e-6 * max(1.0, float(seen["loc"].abs().max()))
    assert float((seen["fw"] - seen["w"]).abs().max()) <= 1e-7
    assert got.shape == want.shape == (bs, A, 512)
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    if f16:   # same launch on the widened rows: the same numbers (the compiler may contract the two loops differently)
        with torch.no_grad(), routes.override(dfa_f16_tokens=False):
            wide = dfa(feat, anchor, emb, fm, m)
        assert float((wide - got).abs().max()) <= 2e-6 * max(1.0, float(got.abs().max()))


@pytest.mark.parametrize("split", [False, True], ids=["fp32_mfma", "split_fp16"])
@pytest.mark.parametrize("bs,nq,nk,scale", [(1, 900, 600, 1.0), (2, 77, 77, 1.0), (1, 33, 1, 1.0), (1, 900, 900, 4.0)])
def test_attention_f32_vs_float64(bs, nq, nk, scale, split):
    """Both instantiations of the attention core (exact-fp32 matrix instruction; FP16 matrix cores with split operands,
    what a frame runs) against float64, the same bound; scale 4: operands of magnitude ~16 and logits in the hundreds
    (sharp softmax), where a 22-bit operand split must still hold."""
    rs = np.random.RandomState(8)
    q = torch.from_numpy(rs.standard_normal((bs, nq, 512)).astype(np.float32)) * scale
    k = torch.from_numpy(rs.standard_normal((bs, nk, 512)).astype(np.float32)) * scale
    v = torch.from_numpy(rs.standard_normal((bs, nk, 512)).astype(np.float32)) * scale
    got = _ops().attention_f32(q.cuda(), k.cuda(), v.cuda(), 8, split=split).cpu()
    qd, kd, vd = (t.double().reshape(bs, -1, 8, 64).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd).transpose(1, 2).reshape(bs, nq, 512)
    err = float((got.double() - want).abs().max())
    assert err < 2e-5 * max(1.0, float(want.abs().max())), err


@pytest.mark.parametrize("split", [False, True], ids=["fp32_mfma", "split_fp16"])
def test_attention_f32_grouped_with_pads_and_strided_views(split):
    """Camera-grouped form against the reference's formulation (dense scores + additive -inf block
    mask + nan_to_num, group_attn.py:104-131), with capacity pads (query_cam = -1), an empty group,
    and q/k passed as strided halves of one fused projection buffer."""
    rs = np.random.RandomState(9)
    bs, n = 2, 150
    bounds = [0, 40, 40, 77, 100, 131, 140]  # group 1 empty; slots 140..149 are capacity pads
    cam = torch.full((n,), -1, dtype=torch.int32)
    for c in range(6):
        cam[bounds[c]:bounds[c + 1]] = c
    qk = torch.from_numpy(rs.standard_normal((bs, n, 1024)).astype(np.float32))
    v = torch.from_numpy(rs.standard_normal((bs, n, 512)).astype(np.float32))
    qk_d = qk.cuda()
    got = _ops().attention_f32(qk_d[..., :512], qk_d[..., 512:], v.cuda(), 8, cam.cuda(),
                               torch.tensor(bounds, dtype=torch.int32).cuda(), split=split).cpu()
    mask = torch.full((n, n), float("-inf"), dtype=torch.float64)
    for c in range(6):
        mask[bounds[c]:bounds[c + 1], bounds[c]:bounds[c + 1]] = 0
    qd, kd, vd = (t.double().reshape(bs, n, 8, 64).transpose(1, 2) for t in (qk[..., :512], qk[..., 512:], v))
    want = torch.nan_to_num(torch.softmax(qd @ kd.transpose(-1, -2) / 8.0 + mask, -1)) @ vd
    want = want.transpose(1, 2).reshape(bs, n, 512)
    assert float((got.double() - want).abs().max()) < 2e-5
    assert float(got[:, 140:].abs().max()) == 0.0


def _pack_split_halfs(x):
    """fp32 -> the 32-bit words simpb_attention_split_halfs reads (what csrc/gemm.hip writes with out_fmt =
    SIMPB_GEMM_OUT_SPLIT_HALFS): low 16 bits half(x), high 16 bits half((x - half(x)) * 2^11), fp32-typed storage."""
    hi = x.half()
    lo = ((x - hi.float()) * 2048.0).half()
    word = (hi.view(torch.int16).to(torch.int32) & 0xFFFF) | (lo.view(torch.int16).to(torch.int32) << 16)
    return word.view(torch.float32)


@pytest.mark.parametrize("nq,nk,scale", [(900, 900, 1.0), (900, 600, 1.0), (77, 45, 1.0), (900, 900, 4.0), (33, 1, 1.0)])
def test_attention_split_halfs_vs_float64(nq, nk, scale):
    """What a frame runs (routes.attention_split_fp16): the eight-wave kernel on operands the producer already split,
    softmax scale folded into q, strided views of one projection buffer; against float64, the exact kernel's bound."""
    rs = np.random.RandomState(18)
    n = max(nq, nk)
    buf = torch.from_numpy(rs.standard_normal((2, n, 1536)).astype(np.float32)) * scale
    q, k, v = buf[:, :nq, :512], buf[:, :nk, 512:1024], buf[:, :nk, 1024:]
    packed = torch.cat([_pack_split_halfs(buf[..., :512] * 0.125), _pack_split_halfs(buf[..., 512:])], -1).cuda()
    got = _ops().attention_f32(packed[:, :nq, :512], packed[:, :nk, 512:1024], packed[:, :nk, 1024:], 8, split=2).cpu()
    qd, kd, vd = (t.double().reshape(2, -1, 8, 64).transpose(1, 2) for t in (q, k, v))
    want = (torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd).transpose(1, 2).reshape(2, nq, 512)
    err = float((got.double() - want).abs().max())
    assert err < 2e-5 * max(1.0, float(want.abs().max())), err


def test_attention_split_halfs_grouped_with_pads():
    """... camera-grouped (an empty group, capacity slots, a partial last tile per group) against the reference's
    formulation (dense scores + additive -inf block mask + nan_to_num, group_attn.py:104-131) in float64."""
    rs = np.random.RandomState(19)
    bs, n = 2, 150
    bounds = [0, 40, 40, 77, 100, 131, 140]
    cam = torch.full((n,), -1, dtype=torch.int32)
    for c in range(6):
        cam[bounds[c]:bounds[c + 1]] = c
    buf = torch.from_numpy(rs.standard_normal((bs, n, 1536)).astype(np.float32))
    packed = torch.cat([_pack_split_halfs(buf[..., :512] * 0.125), _pack_split_halfs(buf[..., 512:])], -1).cuda()
    got = _ops().attention_f32(packed[..., :512], packed[..., 512:1024], packed[..., 1024:], 8, cam.cuda(),
                               torch.tensor(bounds, dtype=torch.int32).cuda(), split=2).cpu()
    mask = torch.full((n, n), float("-inf"), dtype=torch.float64)
    for c in range(6):
        mask[bounds[c]:bounds[c + 1], bounds[c]:bounds[c + 1]] = 0
    qd, kd, vd = (t.double().reshape(bs, n, 8, 64).transpose(1, 2) for t in (buf[..., :512], buf[..., 512:1024], buf[..., 1024:]))
    want = torch.nan_to_num(torch.softmax(qd @ kd.transpose(-1, -2) / 8.0 + mask, -1)) @ vd
    want = want.transpose(1, 2).reshape(bs, n, 512)
    assert float((got.double() - want).abs().max()) < 2e-5
    assert float(got[:, 140:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_format_tokens_vs_feature_maps_format(dtype):
    """One-pass token format against feature_maps_format (itself pinned by ops.npz:fmt.*)."""
    ops = _ops()
    bs, cams, c = 2, 6, 16
    shapes = [(8, 22), (4, 11), (2, 6), (1, 3)]
    maps = [torch.from_numpy(synth.randn(f"fmt2.l{l}", (bs * cams, c, h, w))).to(dtype).cuda() for l, (h, w) in enumerate(shapes)]
    want = ops.feature_maps_format([m.float().reshape(bs, cams, c, *m.shape[-2:]) for m in maps])
    got = ops.format_tokens([m.contiguous(memory_format=torch.channels_last) for m in maps], bs, cams)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])


@pytest.mark.parametrize("shape", [
    dict(bs=2, A=23, P=13, K=6, L=4, G=8, C=256, maps=[(16, 44), (8, 22), (4, 11), (2, 6)]),  # shipped layout
    dict(bs=1, A=7, P=3, K=2, L=2, G=2, C=128, maps=[(5, 7), (3, 2)]),                       # 64 channels per group
])
def test_daf_backward_vs_autograd_of_oracle(shape):
    """Gradients of the HIP operator against torch autograd through the oracle's forward
    (deformable_aggregation_cuda.cu:62-126,190-262 is the analytic form of the same derivative)."""
    R = _oracle()
    s = shape
    rs = np.random.RandomState(11)
    maps = [torch.from_numpy(rs.standard_normal((s["bs"], s["K"], s["C"], h, w)).astype(np.float32)) for h, w in s["maps"]]
    col, ss, ssi = R.feature_maps_format(maps)
    loc = torch.from_numpy(rs.uniform(-0.2, 1.2, (s["bs"], s["A"], s["P"], s["K"], 2)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(0, 1, (s["bs"], s["A"], s["P"], s["K"], s["L"], s["G"])).astype(np.float32))
    gout = torch.from_numpy(rs.standard_normal((s["bs"], s["A"], s["C"])).astype(np.float32))
    c1, l1, w1 = col.clone().requires_grad_(), loc.clone().requires_grad_(), w.clone().requires_grad_()
    R.deformable_aggregation(c1, ss.int(), ssi.int(), l1, w1).backward(gout)
    c2, l2, w2 = (t.clone().cuda().requires_grad_() for t in (col, loc, w))
    out = _ops().deformable_aggregation_function(c2, ss.int().cuda(), ssi.int().cuda(), l2, w2)
    out.backward(gout.cuda())
    for name, got, want in (("feat", c2.grad, c1.grad), ("loc", l2.grad, l1.grad), ("weights", w2.grad, w1.grad)):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got.cpu() - want).abs().max()) <= 2e-4 * scale, name


def test_msda_grouped_backward_vs_autograd_of_oracle():
    """[parity unpinned: the sampler is mmcv's] gradients of the grouped HIP sampler against torch
    autograd through the oracle's grid_sample formulation, per camera group."""
    R = _oracle()
    cfg = dict(bs=2, nq=41, heads=8, ch=32, shapes=[(8, 22), (4, 11), (2, 6), (1, 3)], pts=4, ncam=6)
    value, ss, lsi, loc, aw, groups = _msda_inputs(seed=13, **cfg)
    gout = torch.from_numpy(np.random.RandomState(14).standard_normal((cfg["bs"], cfg["nq"], 256)).astype(np.float32))
    v1, l1, a1 = value.clone().requires_grad_(), loc.clone().requires_grad_(), aw.clone().requires_grad_()
    outs = [R.ms_deform_attn(v1[:, i].contiguous(), ss, l1[:, s:e].contiguous(), a1[:, s:e].contiguous())
            for i, (s, e) in enumerate(groups) if e > s]
    torch.cat(outs, dim=1).backward(gout)
    ops = _ops()
    qcam = ops.query_cam_from_groups(groups, cfg["nq"], "cuda")
    v2, l2, a2 = (t.clone().cuda().requires_grad_() for t in (value, loc, aw))
    ops.ms_deform_attn_grouped(v2, ss.cuda(), lsi.cuda(), l2, a2, qcam).backward(gout.cuda())
    for name, got, want in (("value", v2.grad, v1.grad), ("loc", l2.grad, l1.grad), ("attn", a2.grad, a1.grad)):
        scale = max(float(want.abs().max()), 1.0)
        assert float((got.cpu() - want).abs().max()) <= 2e-4 * scale, name


def test_fused_backbone_epilogue_matches_plain_fp16_backbone():
    """ResNet50+FPN fp16 with folded BN: the one-kernel conv epilogue (bias [+residual] [+ReLU]) against
    the same folded network run with PyTorch's add_/add/relu_. Both are fp16; the fused epilogue rounds
    once instead of up to three times, hence the fp16-level tolerance."""
    from simpb_amd import configs, plugin
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    model = model.cuda().fuse_conv_bn().half_backbone()
    img = synth.images(1, 0, (352, 128)).cuda()
    with torch.no_grad():
        got = model.extract_feat(img)[0]
        model.img_backbone.fused_epilogue = False
        for name in model.img_backbone.res_layers:
            for blk in getattr(model.img_backbone, name):
                blk.fused_epilogue = False
        want = model.extract_feat(img)[0]
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-2 * scale
    assert float((got - want).abs().mean()) <= 2e-3 * scale


def test_mlp_chain_kernel_variants_agree():
    """The three kernels behind simpb_mlp_chain_forward agree: 4-row workgroups on the 4x4 matrix blocks with k4-packed
    weights (shipped), 16-row workgroups on 16x16 tiles with the weights as stored, VALU on transposed weights; on a row
    count that is not a multiple of either tile, through a refinement chain (256-wide layers, LayerNorms, 11-wide head)
    and through the 3D anchor encoder (3- and 2-wide first layers, 128/32/32/64-wide branches)."""
    from simpb_amd.plugin import fused
    from simpb_amd.plugin.detection3d import SparseBox3DEncoder, SparseBox3DRefinementModule
    ref3 = SparseBox3DRefinementModule(embed_dims=256, num_cls=10, refine_yaw=True, with_quality_estimation=True).cuda()
    synth.load_procedural(ref3, seed=7)
    x = torch.randn(1, 333, 256, device="cuda")
    e = torch.randn(1, 333, 256, device="cuda")
    enc = SparseBox3DEncoder(embed_dims=[128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4).cuda()
    synth.load_procedural(enc, seed=9)
    anchor = torch.randn(1, 333, 11, device="cuda")
    from simpb_amd.plugin import routes
    outs = {}
    for name, (r4, tr) in dict(rows4=(True, False), rows16=(False, False), valu=(False, True)).items():
        with routes.override(chain_rows4=r4, chain_transposed=tr), torch.no_grad():
            outs[name] = (fused.chain_forward(ref3.layers, x, e), enc(anchor))
    for name in ("rows16", "valu"):
        for got, want in zip(outs["rows4"], outs[name]):
            assert got.shape == want.shape
            assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), name


def test_mlp_chain_32_row_kernel_equals_the_4_row_kernel():
    """Launches with thousands of rows (a batch of camera streams per launch) take mlp_chain_r32_kernel (32x32 matrix tiles,
    fragment-packed weights); it must agree with the shipped 4-row kernel on the same rows -- 3D anchor encoder (2-/3-wide
    first layers on the vector units, 128/32/32/64-wide branches = 4/1/1/2 column tiles), both refinement
    heads with their post stages (11-, 10-, 4-wide last layers: one partly filled tile), the LayerNorm of the decoder's
    `norm` operator inside the launch, capacity rows (m_live) as zeros, a row count that is no multiple of 32 -- and with
    the PyTorch modules on the CPU."""
    from simpb_amd.plugin import fused
    from simpb_amd.plugin.detection2d import SparseBox2DEncoder, SparseBox2DRefinementModule
    from simpb_amd.plugin.detection3d import SparseBox3DEncoder, SparseBox3DRefinementModule
    n = 4171
    torch.manual_seed(1)
    enc = SparseBox3DEncoder(embed_dims=[128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4)
    enc2 = SparseBox2DEncoder(embed_dims=256, with_sin_embed=True, in_loops=1, out_loops=2)
    ref3 = SparseBox3DRefinementModule(embed_dims=256, num_cls=10, refine_yaw=True, with_quality_estimation=True)
    ref2 = SparseBox2DRefinementModule(embed_dims=256, num_cls=10, with_alpha_branch=True)
    norm = torch.nn.LayerNorm(256)
    for seed, m in enumerate((enc, enc2, ref3, ref2, norm)):
        synth.load_procedural(m, seed=11 + seed)
    box, pts = torch.randn(1, n, 11), torch.rand(1, n, 2) * 1.2 - 0.1
    f, e, a2 = torch.randn(1, n, 256), torch.randn(1, n, 256), torch.rand(1, n, 2)
    dt = torch.tensor([0.5])
    live = torch.tensor([3000], dtype=torch.int32, device="cuda")
    with torch.no_grad():
        want = dict(enc=torch.cat([enc.pos_fc(box[..., 0:3]), enc.size_fc(box[..., 3:6]), enc.yaw_fc(box[..., 6:8]),
                                   enc.vel_fc(box[..., 8:11])], -1), enc2=enc2(pts), ref3=ref3(norm(f), box, e, dt, True),
                    ref2=ref2(norm(f), a2, e))
    mods = [m.cuda() for m in (enc, enc2, ref3, ref2, norm)]

    def run():
        with torch.no_grad():
            out = dict(enc=enc(box.cuda()), enc2=enc2(pts.cuda(), m_live=live),
                       ref3=ref3(f.cuda(), box.cuda(), e.cuda(), dt.cuda(), True, norm=norm),
                       ref2=ref2(f.cuda(), a2.cuda(), e.cuda(), m_live=live, norm=norm))
            out["norm3"], out["norm2"] = None, ref2.norm_out
            return out

    wide = run()
    keep = fused.WIDE_ROWS
    try:
        fused.WIDE_ROWS = 1 << 30
        narrow = run()
    finally:
        fused.WIDE_ROWS = keep

    def leaves(x):
        return [t for t in (x if isinstance(x, (tuple, list)) else [x]) if torch.is_tensor(t)]

    for k in ("enc", "enc2", "ref3", "ref2", "norm2"):
        rows = 3000 if k in ("enc2", "ref2", "norm2") else n   # (rows past m_live are capacity slots: zeros or whatever the tile computed)
        for g_, w_ in zip(leaves(wide[k]), leaves(narrow[k])):
            assert g_.shape == w_.shape
            assert float((g_[:, :rows] - w_[:, :rows]).abs().max()) < 3e-5 * max(1.0, float(w_[:, :rows].abs().max())), k
    for k in ("enc", "ref3"):
        for g_, w_ in zip(leaves(wide[k]), leaves(want[k])):
            assert float((g_.cpu() - w_).abs().max()) < 1e-4 * max(1.0, float(w_.abs().max())), k
    for k in ("enc2", "ref2"):   # rows past m_live are capacity slots
        for g_, w_ in zip(leaves(wide[k]), leaves(want[k])):
            assert float((g_.cpu()[:, :3000] - w_[:, :3000]).abs().max()) < 1e-4 * max(1.0, float(w_.abs().max())), k
    assert float(wide["enc2"][:, 3008:].abs().max()) == 0.0   # workgroups whose 32 rows are all capacity slots write zeros


def test_aggregate_with_alpha_in_launch_equals_two_launches():
    """ReWeight.alpha folded into the 2D -> 3D aggregation launch (csrc/alloc.hip) against the row-dot launch followed by the
    aggregation on its output, and against the definition (aggregation.py:23-35) in float64."""
    from simpb_amd.plugin import dense
    from simpb_amd.plugin.allocation import aggregate_2d_to_3d
    g = torch.Generator().manual_seed(3)
    bs, A, cams, N2, C = 2, 50, 6, 130, 256
    a2q = torch.full((bs, A, cams), -1, dtype=torch.int32)
    for b in range(bs):   # every slot belongs to at most one (anchor, cam)
        perm = torch.randperm(N2, generator=g)[:100]
        pos = torch.randperm(A * cams, generator=g)[:100]
        a2q[b].view(-1)[pos] = perm.to(torch.int32)
    q3d, pos3d = torch.randn(bs, A, C, generator=g), torch.randn(bs, A, C, generator=g)
    q2d, pos2d = torch.randn(bs, N2, C, generator=g), torch.randn(bs, N2, C, generator=g)
    hidden = torch.relu(torch.randn(bs, N2, C, generator=g))
    fc = torch.nn.Linear(C, 1)
    with torch.no_grad():
        fc.weight.copy_(torch.randn(1, C, generator=g) * 0.1)
        fc.bias.fill_(0.3)
    fc = fc.cuda()
    args = [t.cuda() for t in (q3d, pos3d, q2d, pos2d)]
    with torch.no_grad():
        alpha = dense.rowdot_sigmoid(hidden.cuda(), fc.weight, fc.bias)
        want = aggregate_2d_to_3d(*args, alpha, a2q.cuda())
        got = aggregate_2d_to_3d(*args, None, a2q.cuda(), hidden=hidden.cuda(), alpha_fc=fc)
    for x, y in zip(got, want):
        assert float((x - y).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))
    al = torch.sigmoid(hidden.double() @ fc.weight.detach().cpu().double().t() + 0.3)[..., 0]
    ref = q3d.double().clone()
    for b in range(bs):
        for a in range(A):
            s = a2q[b, a][a2q[b, a] >= 0].long()
            if len(s):
                ref[b, a] += (al[b, s, None] * q2d[b, s].double()).sum(0) / al[b, s].sum().clamp(min=1e-5)
    assert float((got[0].cpu().double() - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize("live", [None, 301])
def test_norm_inside_the_refinement_launch_equals_norm_then_head(live):
    """The decoder's `norm` operator applied by the refinement head's own chain launch (leading LayerNorm stage of the 4-row
    chain kernel, csrc/mlp_chain.hip) against the LayerNorm launch followed by the head: same box / class / quality (alpha)
    outputs and the same operator output, 3D and 2D heads, with and without capacity rows."""
    from simpb_amd.plugin import dense
    from simpb_amd.plugin.detection2d import SparseBox2DRefinementModule
    from simpb_amd.plugin.detection3d import SparseBox3DRefinementModule
    g = torch.Generator().manual_seed(11)
    n = 333
    x = (torch.randn(1, n, 256, generator=g) * 1.5 + 0.2).cuda()
    e = torch.randn(1, n, 256, generator=g).cuda()
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(256, generator=g))
        ln.bias.copy_(0.1 * torch.randn(256, generator=g))
    ln = ln.cuda()
    ml = torch.tensor([live], dtype=torch.int32, device="cuda") if live is not None else None
    rows = live if live is not None else n
    with torch.no_grad():
        if live is None:
            ref3 = SparseBox3DRefinementModule(embed_dims=256, num_cls=10, refine_yaw=True, with_quality_estimation=True).cuda()
            synth.load_procedural(ref3, seed=7)
            anchor = torch.randn(1, n, 11, generator=g).cuda()
            dt = torch.tensor([0.5], device="cuda")
            want = ref3(dense.layernorm(x, ln), anchor, e, time_interval=dt, return_cls=True)
            got = ref3(x, anchor, e, time_interval=dt, return_cls=True, norm=ln)
            for a, b in zip(got, want):
                assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
            assert float((ref3.norm_out - dense.layernorm(x, ln)).abs().max()) <= 1e-5
        ref2 = SparseBox2DRefinementModule(embed_dims=256, with_alpha_branch=True).cuda()
        synth.load_procedural(ref2, seed=3)
        a2 = torch.rand(1, n, 2, generator=g).cuda()
        xn = dense.layernorm(x, ln, m_live=ml)
        want = ref2(xn, a2, e, m_live=ml)
        got = ref2(x, a2, e, m_live=ml, norm=ln)
        for a, b in zip(got, want):
            if b is not None:
                assert float((a[:, :rows] - b[:, :rows]).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
        assert float((ref2.norm_out - xn).abs().max()) <= 1e-5
        if live is not None:
            assert float(ref2.norm_out[:, live:].abs().max()) == 0.0


@pytest.mark.parametrize("bs,threshold", [(8, None), (8, 0.3), (3, 0.3), (17, None)])
def test_bank_cache_one_workgroup_per_stream_equals_the_serial_walk(bs, threshold):
    """simpb_bank_cache_streams (csrc/bank.hip bank_cache_streams_kernel: one workgroup per stream, fresh ids offset by the
    counts of the streams in front, one meeting inside the launch) against the serial kernel on the same state: kept
    confidences / rows / ids, the ids of all instances and prev_id, bit for bit, over three consecutive commits (the
    second and third start from ids the first left, with a partly tracked bank)."""
    from simpb_amd import _lib
    from simpb_amd.plugin.ops import _ptr, _stream
    lib = _lib.lib()
    g = torch.Generator().manual_seed(100 + bs)
    A, T, C, E = 900, 600, 10, 256

    def fresh_state():
        return dict(conf=torch.rand(bs, T, generator=g).cuda(), cf=torch.zeros(bs, T, E).cuda(), ca=torch.zeros(bs, T, 11).cuda(),
                    iid=torch.full((bs, A), -1, dtype=torch.long).cuda(), prev=torch.zeros((), dtype=torch.long).cuda())

    base = fresh_state()
    states = [dict((k, v.clone()) for k, v in base.items()) for _ in range(2)]
    sync = torch.zeros(2, dtype=torch.int32, device="cuda")
    for step in range(3):
        feat = torch.randn(bs, A, E, generator=g).cuda()
        anchor = torch.randn(bs, A, 11, generator=g).cuda()
        cls = (torch.randn(bs, A, C, generator=g) * 2).cuda()
        outs = []
        for which, st in enumerate(states):
            ids_out = torch.empty(bs, A, dtype=torch.long, device="cuda")
            scratch = torch.empty(bs, T, dtype=torch.int32, device="cuda")
            _lib.check(lib.simpb_bank_cache_streams(
                _ptr(st["conf"]), _ptr(st["cf"]), _ptr(st["ca"]), _ptr(st["iid"]), _ptr(st["prev"]), _ptr(ids_out), _ptr(scratch),
                _ptr(feat), _ptr(anchor), _ptr(cls), bs, A, C, T, E, 1 if step else 0, 0.6, 0 if threshold is None else 1,
                0.0 if threshold is None else threshold, None, 0, None, _ptr(sync) if which == 1 else None, _stream()), "bank_cache")
            outs.append(ids_out)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), step
        for k in ("conf", "cf", "ca", "iid", "prev"):
            assert torch.equal(states[0][k], states[1][k]), (step, k)
        assert int(states[1]["prev"]) > 0 and int(sync[1]) == step + 1 and int(sync[0]) == (step + 1) * bs


def test_bank_cache_per_stream_writes_nothing_while_a_hold_flag_is_set():
    """The frame-end commit holds back when an overflow flag (or the sticky word of the frame before) is set -- the frame is
    going to be re-run on the state it found (runner.py). The per-stream kernel must then leave the state, prev_id AND its
    own meeting counters untouched (every workgroup returns in front of the meeting), and commit normally afterwards."""
    from simpb_amd import _lib
    from simpb_amd.plugin.ops import _ptr, _stream
    lib = _lib.lib()
    g = torch.Generator().manual_seed(7)
    bs, A, T, C, E = 4, 900, 600, 10, 256
    st = dict(conf=torch.rand(bs, T, generator=g).cuda(), cf=torch.randn(bs, T, E, generator=g).cuda(), ca=torch.randn(bs, T, 11, generator=g).cuda(),
              iid=torch.randint(-1, 50, (bs, A), generator=g).cuda(), prev=torch.tensor(1234).cuda())
    before = {k: v.clone() for k, v in st.items()}
    sync = torch.zeros(2, dtype=torch.int32, device="cuda")
    feat, anchor = torch.randn(bs, A, E, generator=g).cuda(), torch.randn(bs, A, 11, generator=g).cuda()
    cls = torch.randn(bs, A, C, generator=g).cuda()
    ids_out = torch.full((bs, A), -7, dtype=torch.long, device="cuda")
    scratch = torch.empty(bs, T, dtype=torch.int32, device="cuda")
    sticky = torch.zeros(1, dtype=torch.int32, device="cuda")

    def commit(hold):
        _lib.check(lib.simpb_bank_cache_streams(
            _ptr(st["conf"]), _ptr(st["cf"]), _ptr(st["ca"]), _ptr(st["iid"]), _ptr(st["prev"]), _ptr(ids_out), _ptr(scratch),
            _ptr(feat), _ptr(anchor), _ptr(cls), bs, A, C, T, E, 1, 0.6, 0, 0.0, _ptr(hold), hold.numel(), _ptr(sticky), _ptr(sync),
            _stream()), "bank_cache")
        torch.cuda.synchronize()

    commit(torch.tensor([0, 1, 0], dtype=torch.int32, device="cuda"))
    for k in st:
        assert torch.equal(st[k], before[k]), k
    assert int(sticky) == 1 and sync.tolist() == [0, 0] and bool((ids_out == -7).all())
    sticky.zero_()
    commit(torch.zeros(3, dtype=torch.int32, device="cuda"))
    assert int(sticky) == 0 and sync.tolist() == [bs, 1] and int(st["prev"]) > 1234 and not torch.equal(st["conf"], before["conf"])
    assert bool((ids_out >= 0).all())   # no threshold: every instance has an id after a commit


@pytest.mark.parametrize("rows_in,rows_out,kept", [(1536, 1800, 377), (700, 64, 0), (300, 64, 64), (2304, 96, 130)])
def test_record2d_compact_vs_torch(rows_in, rows_out, kept):
    """csrc/decode.hip record2d_compact_kernel (the exchange's 2D payload) against the torch statement of the same
    compaction (simpb_amd.dist.compact_record2d on CPU tensors): kept rows in slot order, pad rows behind; no kept rows at
    all; exactly full; more kept rows than the output holds (clipped); a strided output (streams of a wider send buffer)."""
    from simpb_amd.dist import compact_record2d
    g = torch.Generator().manual_seed(rows_in + kept)
    bs = 3
    rec = torch.rand(bs, rows_in, 8, generator=g)
    rec[..., 6:8] = -1.0
    for b in range(bs):
        idx = torch.randperm(rows_in, generator=g)[:kept]
        rec[b, idx, 6] = torch.randint(0, 300, (kept,), generator=g).float()
        rec[b, idx, 7] = torch.randint(0, 6, (kept,), generator=g).float()
        rec[b, torch.randperm(rows_in, generator=g)[:50], 7] = 2.0   # a camera but no kept box: not sent
    want = compact_record2d(rec.clone(), rows_out)
    got = compact_record2d(rec.cuda(), rows_out).cpu()
    assert torch.equal(got, want)
    wide = torch.full((bs, 40 + rows_out * 8), 9.0, device="cuda")   # the exchange's send buffer: 3D part in front
    compact_record2d(rec.cuda(), rows_out, out=wide[:, 40:].unflatten(1, (rows_out, 8)))
    assert torch.equal(wide[:, 40:].cpu().reshape(bs, rows_out, 8), want) and bool((wide[:, :40] == 9.0).all())
