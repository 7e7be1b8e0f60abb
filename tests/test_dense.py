"""csrc/gemm.hip + plugin/dense.py: the grouped segment-input GEMM, the segmented LayerNorm and the
weight folds. CPU part: the folds are exact algebra (checked in float64 against the unfused formulas
of simpb_head.py:298-310 and blocks.py:384-393) and the row-view logic. GPU part: the kernels against
float64 matmuls; the fused attention / FFN blocks against the unfused module route on the GPU.
Tolerance for fp32 GEMMs: 2e-5 * max|ref| (both sides fp32-exact products, different summation order)."""
import numpy as np
import pytest

from simpb_amd import synth
import torch
import torch.nn as nn
import torch.nn.functional as F

from simpb_amd.plugin import dense


def _mha(e=512, h=8, seed=0):
    torch.manual_seed(seed)
    attn = nn.MultiheadAttention(e, h)
    with torch.no_grad():
        attn.in_proj_bias.normal_(0, 0.1)
        attn.out_proj.bias.normal_(0, 0.1)
    return attn


def test_rows2d_views():
    base = torch.zeros(2, 10, 1536)
    t, rows, ld = dense.rows2d(base[..., 512:1024])
    assert t.data_ptr() == base[..., 512:1024].data_ptr() and rows == 20 and ld == 1536
    ex = torch.zeros(10, 256)[None].expand(1, -1, -1)
    t, rows, ld = dense.rows2d(ex)
    assert rows == 10 and ld == 256 and t.data_ptr() == ex.data_ptr()
    ex2 = torch.zeros(10, 256)[None].expand(2, -1, -1)   # stride-0 batch: must be copied
    t, rows, ld = dense.rows2d(ex2)
    assert rows == 20 and ld == 256 and t.is_contiguous()
    odd = torch.zeros(4, 10, 256)[:, :5]                  # non-uniform row stride
    t, rows, ld = dense.rows2d(odd)
    assert rows == 20 and ld == 256 and t.is_contiguous()


def test_fold_mha_matches_unfused_formula():
    """graph_model (simpb_head.py:298-310): fc_after(cat(f, p) + out_proj(o)), v = in_proj_v(fc_before(f))."""
    attn = _mha()
    c, e = 256, 512
    torch.manual_seed(1)
    pre, post = nn.Linear(c, e, bias=False), nn.Linear(e, c, bias=False)
    f, p, o = torch.randn(7, c).double(), torch.randn(7, c).double(), torch.randn(7, e).double()
    w, b = dense.fold_mha_in(attn, pre, "qkv")
    got = torch.cat([f, p], 1) @ w.double().t() + b.double()
    x = torch.cat([f, p], 1)
    W, B = attn.in_proj_weight.double(), attn.in_proj_bias.double()
    want = torch.cat([x @ W[:e].t() + B[:e], x @ W[e:2 * e].t() + B[e:2 * e],
                      (f @ pre.weight.double().t()) @ W[2 * e:].t() + B[2 * e:]], 1)
    assert float((got - want).abs().max()) < 1e-5
    wq, bq = dense.fold_mha_in(attn, None, "q")
    wkv, bkv = dense.fold_mha_in(attn, pre, "kv")
    assert torch.equal(wq, w[:e]) and torch.equal(wkv, w[e:]) and torch.equal(torch.cat([bq, bkv]), b)
    # no fc_before: v = in_proj_v(cat)
    w2, _ = dense.fold_mha_in(attn, None, "qkv")
    assert torch.equal(w2, attn.in_proj_weight.detach())
    wo, bo = dense.fold_mha_out(attn, post)
    got = torch.cat([o, f, p], 1) @ wo.double().t() + bo.double()
    want = (x + o @ attn.out_proj.weight.double().t() + attn.out_proj.bias.double()) @ post.weight.double().t()
    assert float((got - want).abs().max()) < 1e-5


def test_fold_cache_follows_parameter_updates():
    attn = _mha(seed=3)
    post = nn.Linear(512, 256, bias=False)
    w0, _ = dense.fold_mha_out(attn, post)
    assert dense.fold_mha_out(attn, post)[0] is w0          # cached
    with torch.no_grad():
        post.weight.mul_(2.0)                               # in-place update bumps the version
    w1, _ = dense.fold_mha_out(attn, post)
    assert w1 is not w0 and torch.allclose(w1, 2 * w0, rtol=1e-6, atol=1e-7)


def test_fold_ffn_and_sum_input():
    torch.manual_seed(2)
    fc2, idfc, lin = nn.Linear(1024, 256), nn.Linear(512, 256), nn.Linear(256, 416)
    h, x = torch.randn(5, 1024), torch.randn(5, 512)
    w, b = dense.fold_ffn_out(fc2, idfc)
    assert torch.allclose(torch.cat([h, x], 1) @ w.t() + b, fc2(h) + idfc(x), atol=1e-5)
    a, c = torch.randn(5, 256), torch.randn(5, 256)
    assert torch.allclose(torch.cat([a, c], 1) @ dense.fold_sum_input(lin).t() + lin.bias, lin(a + c), atol=1e-5)
    w, b = dense.fold_stack("t", [nn.Linear(256, 8), nn.Linear(256, 4)], copies=2)
    assert w.shape == (12, 512) and b.shape == (12,)


# ---------------------------------------------------------------------------------------- GPU
gpu = pytest.mark.gpu


def _ref(xs, w, b, relu):
    y = torch.cat([x.double().reshape(-1, x.shape[-1]) for x in xs], 1) @ w.double().t()
    if b is not None:
        y = y + b.double()
    return y.clamp_min(0) if relu else y


@gpu
@pytest.mark.parametrize("m,n,ks,relu,bias", [
    (900, 256, [512, 256, 256], False, True),    # attention output fold: 232 tiles of 32x32
    (900, 1536, [256, 256], False, True),        # q|k|v projection: 64-wide tiles
    (900, 1024, [512], True, True),              # FFN fc1 + ReLU
    (900, 256, [1024, 512], False, True),        # FFN output fold, K = 1536
    (900, 18, [256], False, True),               # learnable_fc: N < one tile
    (6, 416, [256], False, False),               # camera logits: M < one tile
    (1536, 384, [256, 256], False, True),        # MSDA offsets|weights
    (33, 100, [64], True, False),                # ragged everything, single chunk
    (65, 70, [64, 64, 64, 64], False, True),     # four segments
])
@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_fp16"])
def test_gemm_vs_float64(m, n, ks, relu, bias, split):
    """split_fp16: the FP16-matrix-core variant (used when every segment is 128-aligned) must meet the same bound."""
    from simpb_amd.plugin import routes
    with routes.override(gemm_split_fp16=split):
        _gemm_vs_float64(m, n, ks, relu, bias)


def _gemm_vs_float64(m, n, ks, relu, bias):
    rs = torch.Generator().manual_seed(m * 7 + n)
    xs = [torch.randn(m, k, generator=rs) for k in ks]
    w = torch.randn(n, sum(ks), generator=rs) / np.sqrt(sum(ks))
    b = torch.randn(n, generator=rs) if bias else None
    got = dense.linear([x.cuda() for x in xs], w.cuda(), b.cuda() if bias else None, relu=relu).cpu()
    want = _ref(xs, w, b, relu)
    assert got.shape == (m, n)
    assert float((got.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@gpu
def test_gemm_grouped_strided_and_m_live():
    """Three problems in one launch; segments and outputs that are column ranges of wider buffers;
    rows >= *m_live come back as zeros and rows below are untouched by the bound."""
    g = torch.Generator().manual_seed(5)
    wide = torch.randn(2, 50, 1536, generator=g).cuda()
    pos = torch.randn(2, 50, 256, generator=g).cuda()
    w0 = (torch.randn(512, 768, generator=g) / 28).cuda()
    w1 = (torch.randn(96, 256, generator=g) / 16).cuda()
    b1 = torch.randn(96, generator=g).cuda()
    big = torch.randn(1, 700, 128, generator=g).cuda()
    w2 = (torch.randn(64, 128, generator=g) / 11).cuda()
    out1 = torch.full((2, 50, 200), 7.0, device="cuda")
    live = torch.tensor([0, 0, 411], dtype=torch.int32, device="cuda")
    y0, y1, y2 = dense.gemm(
        dense.job([wide[..., 512:1024], pos], w0),
        dense.job(pos, w1, b1, relu=True, out=out1[..., 100:196]),
        dense.job(big, w2, m_live=live[2:3]))
    want0 = _ref([wide[..., 512:1024].cpu(), pos.cpu()], w0.cpu(), None, False).reshape(2, 50, 512)
    assert float((y0.cpu().double() - want0).abs().max()) <= 2e-5 * float(want0.abs().max())
    want1 = _ref([pos.cpu()], w1.cpu(), b1.cpu(), True).reshape(2, 50, 96)
    assert y1.data_ptr() == out1[..., 100:196].data_ptr()
    assert float((out1[..., 100:196].cpu().double() - want1).abs().max()) <= 2e-5 * float(want1.abs().max())
    assert bool((out1[..., :100] == 7.0).all()) and bool((out1[..., 196:] == 7.0).all())   # neighbours untouched
    want2 = _ref([big.cpu()], w2.cpu(), None, False).reshape(1, 700, 64)
    assert float((y2[:, :411].cpu().double() - want2[:, :411]).abs().max()) <= 2e-5 * float(want2.abs().max())
    assert bool((y2[:, 411:] == 0).all())


@gpu
@pytest.mark.parametrize("m,n,ks,relu", [
    (2250, 1536, [256, 256], False),      # q|k|v of two streams' 2D sets: 36 x 12 tiles of 64 x 128, partial last row tile
    (3300, 1000, [512], True),            # N not a multiple of the tile: partial column tiles
    (14405, 256, [1024, 512], True),      # a batch of streams: narrow output, many row tiles, two segments, K = 1536
    (2700, 1536, [128, 128, 128, 128], False),   # four segments
])
def test_gemm_wide_tiles_vs_float64(m, n, ks, relu):
    """Launches with >= 400 tiles of 64 x 128 take csrc/gemm.hip gemm_f16x3_wide_kernel (one 32 x 32 tile per wave over the
    whole K, accumulators straight to memory): same bound against float64 as the other forms."""
    _gemm_vs_float64(m, n, ks, relu, True)


@gpu
def test_gemm_wide_tiles_m_live_row_flag_and_split_halfs_output():
    """The wide-tile form with everything a decoder launch asks of it: a device-side live row count (rows past it come back
    as zeros, a tile made of capacity rows only does no work), the flagged second bias, and the (hi, lo) half-pair output
    format of the attention projections -- against the narrow-tile form on the same operands (forced by a launch too
    small for the wide one: the same rows as a 512-row problem)."""
    from simpb_amd.plugin import routes
    g = torch.Generator().manual_seed(31)
    m, n, k, live_n = 3072, 1536, 512, 2260
    x = torch.randn(1, m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / 22).cuda()
    b = torch.randn(n, generator=g).cuda()
    b2 = torch.randn(n, generator=g).cuda()
    flag = (torch.rand(m, generator=g) > 0.5).to(torch.int32).cuda()
    live = torch.tensor([live_n], dtype=torch.int32, device="cuda")
    want = x[0, :live_n].cpu().double() @ w.cpu().double().t() + b.cpu().double() + flag[:live_n].cpu().double()[:, None] * b2.cpu().double()
    got = dense.linear(x, w, b, m_live=live, row_flag=flag, bias2=b2)
    assert float((got[0, :live_n].cpu().double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert bool((got[0, live_n:] == 0).all())
    halfs = dense.linear(x, w, b, m_live=live, row_flag=flag, bias2=b2, split_halfs=True)
    bits = halfs[0, :live_n].view(torch.int32)
    hi = (bits & 0xFFFF).to(torch.int16).view(torch.float16).float()
    lo = ((bits >> 16) & 0xFFFF).to(torch.int16).view(torch.float16).float()
    assert float((hi + lo / 2048.0 - got[0, :live_n]).abs().max()) <= 2e-6 * float(got.abs().max())
    assert bool((halfs[0, live_n:].view(torch.int32) == 0).all())
    # the narrow-tile kernel on a slice small enough to miss the wide form: same numbers up to summation order
    part = dense.linear(x[:, :512], w, b, row_flag=flag[:512].contiguous(), bias2=b2)
    assert float((part[0] - got[0, :512]).abs().max()) <= 2e-5 * float(got.abs().max())


@gpu
def test_gemm_rejects_bad_layouts():
    x = torch.randn(8, 96, device="cuda")      # K not a multiple of 64
    with pytest.raises(RuntimeError):
        dense.linear(x, torch.randn(16, 96, device="cuda"))
    with pytest.raises(RuntimeError):
        dense.linear(torch.randn(8, 64), torch.randn(16, 64))   # CPU tensors: no fallback


@gpu
@pytest.mark.parametrize("ks", [[256], [256, 256], [64]])
def test_layernorm_segments_vs_torch(ks):
    g = torch.Generator().manual_seed(9)
    xs = [(torch.randn(3, 301, k, generator=g) * 3 + 1.5) for k in ks]
    ln = nn.LayerNorm(sum(ks))
    with torch.no_grad():
        ln.weight.normal_(1, 0.2, generator=g)
        ln.bias.normal_(0, 0.2, generator=g)
    want = ln.double()(torch.cat(xs, -1).double())
    ln = ln.float().cuda()
    got = dense.layernorm([x.cuda() for x in xs], ln).cpu()
    assert float((got.double() - want).abs().max()) < 2e-5
    live = torch.tensor([500], dtype=torch.int32, device="cuda")
    got = dense.layernorm([x.cuda() for x in xs], ln, m_live=live).cpu().reshape(-1, sum(ks))
    assert float((got[:500].double() - want.reshape(-1, sum(ks))[:500]).abs().max()) < 2e-5 and bool((got[500:] == 0).all())


@gpu
@pytest.mark.parametrize("halfs", [False, True], ids=["fp32_operands", "split_half_operands"])
def test_fused_graph_attention_matches_unfused_route(halfs):
    """The three-launch block against the module route (cat, fc_before, nn.MultiheadAttention
    arithmetic, + identity, fc_after) on the GPU, for the three call forms of the decoder; with the projections handing
    q / k / v to the attention core as fp32 numbers (exact-fp32 matrix instruction) and as split half pairs with the
    softmax scale folded into the query rows (FP16 matrix cores, routes.attention_split_fp16)."""
    from simpb_amd.plugin import routes
    from simpb_amd.plugin.layers import MultiheadAttention, fused_graph_attention
    torch.manual_seed(11)
    layer = MultiheadAttention(512, 8, batch_first=True, dropout=0.1).cuda().eval()
    with torch.no_grad():
        layer.attn.in_proj_bias.normal_(0, 0.1)
        layer.attn.out_proj.bias.normal_(0, 0.1)
    pre, post = nn.Linear(256, 512, bias=False).cuda(), nn.Linear(512, 256, bias=False).cuda()
    f, p = torch.randn(1, 900, 256, device="cuda"), torch.randn(1, 900, 256, device="cuda")
    tf, tp = torch.randn(1, 600, 256, device="cuda"), torch.randn(1, 600, 256, device="cuda")

    def unfused(query, key, value, qp, kp):
        q = torch.cat([query, qp], -1)
        k = torch.cat([key, kp], -1) if key is not None else None
        v = pre(value) if value is not None else None
        return post(layer(q, k, v))

    with torch.no_grad(), routes.override(attention_split_fp16=halfs):
        for query, key, value, kp in [(f, None, f, None), (f, tf, tf, tp), (f, None, None, None)]:
            got = fused_graph_attention(layer, pre, post, query, p, key, kp, value)
            want = unfused(query, key, value, p, kp)
            assert got is not None
            assert float((got - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))


@gpu
def test_fused_ffn_matches_unfused_route():
    from simpb_amd.plugin.blocks import AsymmetricFFN
    torch.manual_seed(12)
    ffn = AsymmetricFFN(in_channels=512, pre_norm=dict(type="LN"), embed_dims=256, feedforward_channels=1024,
                        num_fcs=2, ffn_drop=0.1, act_cfg=dict(type="ReLU", inplace=True)).cuda().eval()
    a, b = torch.randn(1, 900, 256, device="cuda"), torch.randn(1, 900, 256, device="cuda")
    with torch.no_grad():
        got = ffn(dense.Segments([a, b]))
        x = ffn.pre_norm(torch.cat([a, b], -1))
        want = ffn.identity_fc(x) + ffn.layers(x)
    assert float((got - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))


@gpu
def test_gemm_flagged_extra_bias_is_the_257th_column():
    """ReWeight.reduce over cat(q, is_center) (aggregation.py:19-21) as a 256-wide product + flagged bias."""
    g = torch.Generator().manual_seed(21)
    lin = nn.Linear(257, 256)
    q = torch.randn(1, 700, 256, generator=g)
    flag = (torch.rand(1, 700, generator=g) > 0.5).to(torch.int32)
    want = F.relu(lin.double()(torch.cat([q, flag[..., None].float()], -1).double()))
    lin = lin.float().cuda()
    w_x, w_flag = dense.fold_split_last_column(lin)
    got = dense.linear(q.cuda(), w_x, lin.bias, relu=True, row_flag=flag.cuda().reshape(-1), bias2=w_flag).cpu()
    assert float((got.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    alp = nn.Linear(256, 1)
    want_a = torch.sigmoid(alp.double()(want))
    live = torch.tensor([650], dtype=torch.int32, device="cuda")
    got_a = dense.rowdot_sigmoid(got.cuda(), alp.float().cuda().weight, alp.bias, m_live=live).cpu()
    assert got_a.shape == (1, 700, 1)
    assert float((got_a[:, :650].double() - want_a[:, :650]).abs().max()) < 1e-5 and bool((got_a[:, 650:] == 0).all())


@gpu
def test_anchor_projection_kernel_vs_reference_formula():
    """csrc/rowops.hip against the PyTorch statement of detection3d/blocks.py:248-280 (run on the CPU),
    including the swapped yaw pair."""
    from simpb_amd.plugin.detection3d import SparseBox3DKeyPointsGenerator as K
    g = torch.Generator().manual_seed(22)
    anchor = torch.randn(2, 600, 11, generator=g)
    T = torch.eye(4).repeat(2, 1, 1)
    ang = torch.tensor([0.3, -1.1])
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = ang.cos(), -ang.sin(), ang.sin(), ang.cos()
    T[:, :3, 3] = torch.randn(2, 3, generator=g)
    dt = torch.tensor([0.5, -0.25])
    for ti in (None, [-dt]):
        want = K.anchor_projection(anchor, [T], time_intervals=ti)[0]
        got = K.anchor_projection(anchor.cuda(), [T.cuda()], time_intervals=None if ti is None else [-dt.cuda()])[0].cpu()
        assert float((got - want).abs().max()) < 1e-5


@gpu
def test_msda_prep_kernel_vs_torch_ops():
    """group_attn.py:181-201: softmax over levels x points per head, location = ref + offset / (W, H)."""
    g = torch.Generator().manual_seed(23)
    bs, nq, heads, L, P = 1, 333, 8, 4, 4
    raw = torch.randn(bs, nq, heads * L * P * 3, generator=g)
    ref = torch.rand(bs, nq, 1, 2, generator=g)
    shapes = torch.tensor([[64, 176], [32, 88], [16, 44], [8, 22]])
    off = raw[..., : heads * L * P * 2].reshape(bs, nq, heads, L, P, 2)
    want_w = raw[..., heads * L * P * 2:].reshape(bs, nq, heads, L * P).softmax(-1).reshape(bs, nq, heads, L, P)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    want_loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    from simpb_amd import _lib
    loc = torch.empty(bs, nq, heads, L, P, 2, device="cuda")
    att = torch.empty(bs, nq, heads, L, P, device="cuda")
    live = torch.tensor([300], dtype=torch.int32, device="cuda")
    rawc, refc, sh = raw.cuda(), ref.cuda().reshape(bs * nq, 2), shapes.cuda().long()
    _lib.check(_lib.lib().simpb_msda_prep(loc.data_ptr(), att.data_ptr(), rawc.data_ptr(), raw.shape[-1], refc.data_ptr(), 2,
                                          sh.data_ptr(), bs * nq, heads, L, P, live.data_ptr(), None), "simpb_msda_prep")
    torch.cuda.synchronize()
    assert float((loc.cpu()[:, :300] - want_loc[:, :300]).abs().max()) < 1e-6
    assert float((att.cpu()[:, :300] - want_w[:, :300]).abs().max()) < 1e-6
    assert bool((loc[:, 300:] == 0).all()) and bool((att[:, 300:] == 0).all())


@gpu
def test_chain_post_stages_match_the_module_formulas():
    """The refinement heads with their post stage inside the chain launch (detection3d/blocks.py:133-143,
    detection2d/blocks.py:122-125,144) against the same modules evaluated with plain PyTorch on the CPU."""
    from simpb_amd.plugin.detection2d import SparseBox2DRefinementModule
    from simpb_amd.plugin.detection3d import SparseBox3DRefinementModule
    torch.manual_seed(24)
    r3 = SparseBox3DRefinementModule(embed_dims=256, num_cls=10, refine_yaw=True, with_quality_estimation=True).eval()
    f, e, a = torch.randn(2, 90, 256), torch.randn(2, 90, 256), torch.randn(2, 90, 11)
    dt = torch.tensor([0.5, 0.4])
    with torch.no_grad():
        want = r3(f, a, e, time_interval=dt, return_cls=True)
        got = r3.cuda()(f.cuda(), a.cuda(), e.cuda(), time_interval=dt.cuda(), return_cls=True)
        for w, gt in zip(want, got):
            assert float((gt.cpu() - w).abs().max()) <= 5e-5 * max(1.0, float(w.abs().max()))
        r2 = SparseBox2DRefinementModule(embed_dims=256, num_cls=10, with_alpha_branch=True).eval()
        a2 = torch.rand(2, 90, 2)
        a2[0, 0] = torch.tensor([0.0, 1.0])   # the clamped ends of inverse_sigmoid
        want = r2(f, a2, e)
        got = r2.cuda()(f.cuda(), a2.cuda(), e.cuda())
        for w, gt in zip(want, got):
            if w is not None:
                assert float((gt.cpu() - w).abs().max()) <= 5e-5 * max(1.0, float(w.abs().max()))


@gpu
@pytest.mark.parametrize("bs,n,k", [(1, 900, 600), (2, 900, 300), (1, 300, 300), (3, 37, 5), (1, 2048, 1)])
def test_topk_rows_vs_torch_topk(bs, n, k):
    """Bit-exact values, indices pointing at them, descending order, ties towards the lower index."""
    from simpb_amd.plugin.ops import topk_rows
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(bs, n, generator=g)
    x[0, : min(n, 8)] = 0.5          # ties
    if n > 20:
        x[0, 17] = float("-inf")
        x[0, 3] = -0.0
    want_v, _ = torch.topk(x, k, dim=1)
    got_v, got_i = topk_rows(x.cuda(), k)
    got_v, got_i = got_v.cpu(), got_i.cpu()
    assert got_i.dtype == torch.int64
    assert torch.equal(got_v, want_v)
    assert torch.equal(torch.gather(x, 1, got_i), got_v)
    for b in range(bs):
        assert len(set(got_i[b].tolist())) == k
    tie = (got_v[0, 1:] == got_v[0, :-1])
    assert bool((got_i[0, 1:][tie] > got_i[0, :-1][tie]).all())


@gpu
def test_fused_decode_records_match_the_pytorch_statement():
    """csrc/decode.hip against decode_static_device's PyTorch route (decoder.py:133-175, 23-51) on random heads."""
    from types import SimpleNamespace
    from simpb_amd.plugin import detection3d
    from simpb_amd.plugin.detection3d import SparseBox3DDecoder
    g = torch.Generator().manual_seed(31)
    bs, A, C, N2 = 2, 900, 10, 1536
    cls = [torch.randn(bs, A, C, generator=g).cuda()]
    box = [torch.randn(bs, A, 11, generator=g).cuda()]
    quality = [torch.randn(bs, A, 2, generator=g).cuda()]
    # track ids far above 2^24 (where float32 stops being exact) and above 2^32: they travel as two bit-cast lanes
    ids = (torch.randint(0, 5000, (bs, A), generator=g) + torch.tensor([(1 << 24) + 1, (1 << 40) + 3])[:, None]).cuda()
    cls2d = [torch.randn(bs, N2, C, generator=g).cuda()]
    box2d = [torch.rand(bs, N2, 4, generator=g).cuda()]
    q2a = torch.randint(-1, A, (bs, N2), generator=g).to(torch.int32).cuda()
    cam = torch.randint(-1, 6, (N2,), generator=g).to(torch.int32).cuda()
    alloc = SimpleNamespace(q2a=q2a, query_cam=cam)
    aug = dict(crop=(0, 140, 704, 396), resize=0.44)
    dec = SparseBox3DDecoder(num_output=300)
    from simpb_amd.plugin import routes
    with routes.override(fused_decode=False):
        want3, want2 = dec.decode_static_device(cls, box, ids, quality, cls2d, box2d, alloc, aug)
    got3, got2 = dec.decode_static_device(cls, box, ids, quality, cls2d, box2d, alloc, aug)
    from simpb_amd.dist import lanes_to_ids
    assert got3.shape == want3.shape == (bs, 300, 15) and got2.shape == want2.shape == (bs, N2, 8)
    assert float((got3[..., :13] - want3[..., :13]).abs().max()) < 2e-4      # boxes/scores ~1e-6
    assert float((got2 - want2).abs().max()) < 2e-3      # pixel coordinates up to 1600: 1 ulp = 1.2e-4
    assert torch.equal(got2[..., 5:], want2[..., 5:])    # label, rank, camera exactly
    assert torch.equal(got3[..., 11], want3[..., 11])
    got_ids, want_ids = lanes_to_ids(got3[..., 13:15]), lanes_to_ids(want3[..., 13:15])
    assert torch.equal(got_ids, want_ids) and int(got_ids.min()) > (1 << 24)  # bit-exact int64 ids
    order = torch.argsort(want3[..., 10], dim=1, descending=True)  # (rows are already sorted by score)
    assert set(got_ids[0].tolist()) <= set(ids[0].tolist()) and set(got_ids[1].tolist()) <= set(ids[1].tolist())
    host = SparseBox3DDecoder.decode_static_host(got3.cpu().numpy(), got2.cpu().numpy(), 6)
    assert torch.equal(host[1]["instance_ids"], got_ids[1].cpu()) and host[1]["instance_ids"].dtype == torch.int64


@gpu
def test_fused_bank_matches_the_pytorch_bank_over_a_stream():
    """csrc/bank.hip (get / update / cache + ids on the persistent state) against the PyTorch statement of
    instance_bank.py:79-196 run on the same static state, 4 frames, 2 streams, one stream with a stale gap."""
    from simpb_amd.plugin import instance_bank as ib, routes
    from simpb_amd import synth
    g = torch.Generator().manual_seed(41)
    bs, A, T, C = 2, 900, 600, 256

    def make():
        bank = ib.InstanceBank(num_anchor=A, embed_dims=C, anchor=synth.anchors(A), num_temp_instances=T,
                               anchor_handler=dict(type="SparseBox3DKeyPointsGenerator"), confidence_decay=0.6,
                               feat_grad=False).cuda().eval()
        bank.enable_static(bs, torch.device("cuda"))
        return bank

    frames = []
    for f in range(4):
        Tm = torch.eye(4).repeat(bs, 1, 1)
        Tm[:, :3, 3] = torch.randn(bs, 3, generator=g)
        ang = torch.randn(bs, generator=g) * 0.2
        Tm[:, 0, 0], Tm[:, 0, 1], Tm[:, 1, 0], Tm[:, 1, 1] = ang.cos(), -ang.sin(), ang.sin(), ang.cos()
        dt = torch.tensor([0.5, 3.0 if f == 2 else 0.5])       # stream 1 goes stale once (|dt| > 2)
        frames.append(dict(T=Tm.cuda(), dt=dt.cuda(), feat1=torch.randn(bs, A, C, generator=g).cuda(),
                           anchor1=torch.randn(bs, A, 11, generator=g).cuda(), cls1=torch.randn(bs, A, 10, generator=g).cuda(),
                           feat2=torch.randn(bs, A, C, generator=g).cuda(), anchor2=torch.randn(bs, A, 11, generator=g).cuda(),
                           cls2=torch.randn(bs, A, 10, generator=g).cuda()))

    def run(fused):
        with routes.override(fused_bank=fused):
            return _run(fused)

    def _run(fused):
        bank, out = make(), []
        with torch.no_grad():
            for f, fr in enumerate(frames):
                metas = {"bank_inputs": (fr["T"], fr["dt"])} if f else {}
                _, _, cf, ca, dt = bank.get(bs, metas)
                rec = dict(dt=dt.clone(), ca=None if ca is None else ca.clone())
                feat, anc = bank.update(fr["feat1"], fr["anchor1"], fr["cls1"])
                rec.update(feat=feat.clone(), anc=anc.clone())
                ids = bank.cache_and_assign_ids(fr["feat2"], fr["anchor2"], fr["cls2"], metas={}, threshold=None)
                if ids is None:
                    bank.cache(fr["feat2"], fr["anchor2"], fr["cls2"], metas={})
                    ids = bank.get_instance_id(fr["cls2"], fr["anchor2"], None)
                rec.update(ids=ids.clone(), conf=bank.confidence.clone(), cfeat=bank.cached_feature.clone(),
                           canc=bank.cached_anchor.clone(), kept=bank.instance_id.clone(), prev=int(bank.prev_id))
                out.append(rec)
        return out

    want, got = run(False), run(True)
    for f, (w, gt) in enumerate(zip(want, got)):
        for k in w:
            if w[k] is None:
                assert gt[k] is None
            elif isinstance(w[k], int):
                assert w[k] == gt[k], (f, k)
            elif w[k].dtype == torch.long:
                assert torch.equal(w[k], gt[k]), (f, k)
            else:
                assert float((w[k] - gt[k]).abs().max()) < 1e-5, (f, k)


@gpu
@pytest.mark.parametrize("m,scale", [(89760, 1.0), (1000, 1e-3), (77, 100.0)])
def test_linear_split_fp16_passes_reach_fp32_accuracy(m, scale):
    """csrc/linear_split.hip against float64, same bound as the exact fp32 kernel (2e-5 * max|ref|), on
    small, unit and large magnitudes (the trailing parts are scaled, so small inputs do not underflow)."""
    from simpb_amd.plugin.ops import linear_f32, linear_split
    g = torch.Generator().manual_seed(m)
    x = (torch.randn(m, 256, generator=g) * scale)
    w = torch.randn(256, 256, generator=g) / 16
    b = torch.randn(256, generator=g) * scale
    want = x.double() @ w.double().t() + b.double()
    got = linear_split(x.cuda(), w.cuda(), b.cuda()).cpu()
    exact = linear_f32(x.cuda(), w.cuda(), b.cuda()).cpu()
    bound = 2e-5 * float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= bound
    # and no worse than 4x the exact kernel's own rounding error
    assert float((got.double() - want).abs().max()) <= 4 * float((exact.double() - want).abs().max()) + 1e-7 * scale


@gpu
@pytest.mark.parametrize("m,n,k,scale", [(89760, 256, 256, 1.0), (1000, 256, 256, 1e-3), (77, 40, 192, 100.0), (130, 72, 64, 1.0)])
def test_linear_split_with_half_precision_input(m, n, k, scale):
    """simpb_linear_f16in_split (two passes, x already f16: the camera tokens of the fp16 backbone, simpb.py:63) against
    float64 on the same f16 values, the 2e-5 bound of the exact fp32 kernel, and against the three-pass kernel on the
    widened x (same terms: the trailing part of x is zero; only the summation order differs). Ragged M / N, 1 and 3
    K chunks included."""
    from simpb_amd.plugin.ops import linear_split
    g = torch.Generator().manual_seed(m + n)
    x = (torch.randn(m, k, generator=g) * scale).half()
    w = torch.randn(n, k, generator=g) / 16
    b = torch.randn(n, generator=g) * scale
    want = x.double() @ w.double().t() + b.double()
    got = linear_split(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert got.dtype == torch.float32 and got.shape == (m, n)
    bound = 2e-5 * float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= bound
    if k % 32 == 0:
        three = linear_split(x.float().cuda(), w.cuda(), b.cuda()).cpu()
        assert float((got - three).abs().max()) <= 2e-6 * float(want.abs().max())


@gpu
@pytest.mark.parametrize("cin,cout,h,w,stride,res,relu", [
    (64, 256, 64, 176, 1, True, True),      # layer1 conv3 + residual
    (256, 64, 64, 176, 1, False, True),     # layer1 conv1
    (256, 512, 64, 176, 2, False, False),   # layer2 downsample, stride 2
    (2048, 512, 8, 22, 1, False, True),     # layer4 conv1, small map
    (192, 40, 5, 7, 2, True, True),         # ragged pixels / channels, odd sizes with stride 2, 3 K chunks
])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_conv1x1_kernel_vs_conv2d(cin, cout, h, w, stride, res, relu, variant):
    """csrc/conv1x1.hip against F.conv2d + bias (+ residual) (+ ReLU) evaluated in fp32 on the same fp16 values:
    the kernel accumulates in fp32 and rounds once, so it must sit within one fp16 rounding of that reference."""
    from simpb_amd.plugin.ops import conv1x1_nhwc
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(3, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().cuda()
    b = torch.randn(cout, generator=g).half().cuda()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.randn(3, cout, ho, wo, generator=g).half().cuda().contiguous(memory_format=torch.channels_last) if res else None
    want = F.conv2d(x.float(), wt.float(), b.float(), stride=stride)
    if res:
        want = want + r.float()
    if relu:
        want = want.relu()
    got = conv1x1_nhwc(x, wt, b, r, relu, stride, variant=variant)
    assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float() - want).abs()
    assert float((err - 1e-3 * want.abs()).max()) <= 2e-3   # fp16 output rounding: 2^-11 relative + small absolute


@gpu
@pytest.mark.parametrize("n,c,h,w", [(6, 64, 128, 352), (2, 64, 7, 9), (1, 16, 1, 1), (3, 24, 6, 5)])
def test_stem_epilogue_equals_bias_relu_then_maxpool(n, c, h, w):
    """csrc/bias_act.hip: bias + ReLU + max_pool2d(3, stride 2, padding 1) in one pass against the two-pass route (mmdet
    ResNet.forward: relu(bn1(conv1(x))) -> maxpool, BN folded): bit-equal, odd sizes and 1x1 maps included."""
    from simpb_amd.plugin.ops import bias_act_, bias_relu_maxpool
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, c, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(c, generator=g).half().cuda()
    want = F.max_pool2d(bias_act_(x.clone(), b, None, relu=True), 3, 2, 1)
    got = bias_relu_maxpool(x, b)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)


@gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("cin,cout,h,w,stride,relu", [
    (64, 64, 64, 176, 1, True),       # layer1 conv2
    (128, 128, 64, 176, 2, True),     # layer2.0 conv2, stride 2
    (256, 256, 16, 44, 1, True),      # layer3 conv2
    (512, 512, 8, 22, 1, True),       # layer4 conv2: 72 chunks, small map
    (256, 256, 32, 88, 1, False),     # FPN output convolution (no ReLU)
    (192, 40, 5, 7, 2, True),         # ragged pixels / channels, odd sizes with stride 2, 27 chunks
    (64, 72, 3, 3, 1, False),         # a map smaller than one tile row, second channel block partial
])
def test_conv3x3_kernel_vs_conv2d(cin, cout, h, w, stride, relu, variant):
    """csrc/conv3x3.hip (every tiling) against F.conv2d(padding=1) + bias (+ ReLU) evaluated in fp32 on the same fp16
    values (mmdet ResNet bottleneck conv2 / FPN.fpn_convs after tools/fuse_conv_bn.py:10-48): fp32 accumulate, one
    rounding, so within one fp16 rounding of that reference. Zero padding, stride 2 and image borders are where an
    implicit GEMM goes wrong, hence the small odd shapes; 3 images so that a tile crosses image boundaries."""
    from simpb_amd.plugin.ops import conv3x3_nhwc
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(3, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, generator=g).half().cuda()
    want = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
    if relu:
        want = want.relu()
    got = conv3x3_nhwc(x, wt, b, relu, stride, variant=variant)
    assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float() - want).abs()
    assert float((err - 1e-3 * want.abs()).max()) <= 2e-3


@gpu
@pytest.mark.parametrize("variant", [0, 5, 6])
@pytest.mark.parametrize("cin,cout,h,w,stride,relu", [
    (64, 64, 128, 352, 1, True),      # ResNet101 1408x512 (BASELINE config #4): layer1 conv2, 270 336 pixels
    (128, 128, 128, 352, 2, True),    # layer2.0 conv2, stride 2
    (256, 256, 32, 88, 1, True),      # layer3 conv2 (23 blocks in ResNet101)
    (512, 512, 16, 44, 1, True),      # layer4 conv2
    (256, 256, 128, 352, 1, False),   # FPN output convolution, level 0
    (256, 256, 64, 176, 1, False),    # FPN output convolution, level 1
])
def test_conv3x3_kernel_at_r101_1408x512_shapes(cin, cout, h, w, stride, relu, variant):
    """The same check at the map sizes of the derived ResNet101 1408x512 config with its 6 cameras (the shape-driven
    choice, variant 0, lands on other tilings here than at 704x256) plus the two 128-row staged tilings at these sizes."""
    from simpb_amd.plugin.ops import conv3x3_nhwc
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(6, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half().cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, generator=g).half().cuda()
    want = F.conv2d(x.float(), wt.float(), b.float(), stride=stride, padding=1)
    if relu:
        want = want.relu()
    got = conv3x3_nhwc(x, wt, b, relu, stride, variant=variant)
    assert got.shape == want.shape and got.dtype == torch.float16
    err = (got.float() - want).abs()
    assert float((err - 1e-3 * want.abs()).max()) <= 2e-3


@gpu
@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("cin,cout,h,w,stride,res,relu", [
    (64, 256, 128, 352, 1, True, True),      # ResNet101 1408x512: layer1 conv3 + residual
    (256, 64, 128, 352, 1, False, True),     # layer1 conv1
    (256, 512, 128, 352, 2, False, False),   # layer2 downsample, stride 2
    (1024, 256, 32, 88, 1, False, True),     # layer3 conv1 (x 23)
    (256, 1024, 32, 88, 1, True, True),      # layer3 conv3 + residual
    (2048, 512, 16, 44, 1, False, True),     # layer4 conv1
    (256, 256, 128, 352, 1, False, False),   # FPN lateral, level 0
])
def test_conv1x1_kernel_at_r101_1408x512_shapes(cin, cout, h, w, stride, res, relu, variant):
    """csrc/conv1x1.hip (shape-driven choice and both staged 128-row tilings) at the ResNet101 1408x512 map sizes, 6 cameras."""
    from simpb_amd.plugin.ops import conv1x1_nhwc
    g = torch.Generator().manual_seed(cin + cout + 1)
    x = torch.randn(6, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().cuda()
    b = torch.randn(cout, generator=g).half().cuda()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.randn(6, cout, ho, wo, generator=g).half().cuda().contiguous(memory_format=torch.channels_last) if res else None
    want = F.conv2d(x.float(), wt.float(), b.float(), stride=stride)
    if res:
        want = want + r.float()
    if relu:
        want = want.relu()
    got = conv1x1_nhwc(x, wt, b, r, relu, stride, variant=variant)
    assert got.shape == want.shape and got.dtype == torch.float16
    err = (got.float() - want).abs()
    assert float((err - 1e-3 * want.abs()).max()) <= 2e-3


@gpu
def test_fp16_gpu_backbone_fpn_tokens_vs_fp32_cpu_network():
    """Image -> camera tokens, checked against something other than itself: the fp16 GPU backbone + FPN (own 1x1 / 3x3 / stem
    kernels, BN folded, the FPN's output convolutions writing the token rows) against the SAME folded network evaluated in
    fp32 on the CPU by plain PyTorch (what bench.py's cpu_baseline leg runs; reference caller models/simpb.py:64-91), at
    the shipped R50 704x256 size. fp16 storage of every activation bounds the agreement: max <= 2e-2 x scale, mean <=
    2e-3 x scale (the bound the vendor-convolution cross-check of the neck uses)."""
    from simpb_amd import configs, plugin
    from oracle import simpb_ref as R
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    ref = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(ref)
    ref.fuse_conv_bn()
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    model = model.cuda().fuse_conv_bn().half_backbone()
    img = synth.images(1, 2, (704, 256))
    torch.set_num_threads(16)
    with torch.no_grad():
        feats = ref.img_neck(ref.img_backbone(img.flatten(end_dim=1)))
        want = R.feature_maps_format([x.reshape((1, 6) + x.shape[1:]) for x in feats])
        got = model.extract_feat(img.cuda())
    assert got[0].shape == want[0].shape == (1, 89760, 256)
    assert torch.equal(got[1].cpu().long(), want[1].long()) and torch.equal(got[2].cpu().long(), want[2].long())
    scale = float(want[0].abs().max())
    err = (got[0].float().cpu() - want[0]).abs()
    assert float(err.max()) <= 2e-2 * scale and float(err.mean()) <= 2e-3 * scale, (float(err.max()) / scale, float(err.mean()) / scale)
    half = getattr(got[0], "simpb_f16", None)   # the f16 copy the samplers read holds the same numbers
    assert half is not None and torch.equal(half.float(), got[0])


@gpu
@pytest.mark.parametrize("n,h,w", [(6, 256, 704), (2, 64, 96), (1, 70, 38), (3, 8, 8)])
def test_stem_kernel_vs_conv_bias_relu_maxpool(n, h, w):
    """csrc/stem.hip (cast + 7x7/2 convolution + bias + ReLU + 3x3/2 max-pool, two launches of our own) against
    max_pool2d(relu(conv2d(half(img)) + bias)) evaluated in fp32 on the same f16 values, and against the route it replaces
    (vendor convolution storing an f16 map + csrc/bias_act.hip's fused epilogue). The kernel rounds its fp32 conv sums to f16
    like a stored map, so it sits within one f16 rounding of both; tiles that hang over the map edge, maps that are not a
    multiple of the 8 x 16 pooled tile and odd sizes included."""
    from simpb_amd.plugin.ops import bias_relu_maxpool, stem_conv_pool
    g = torch.Generator().manual_seed(h * 7 + w)
    img = torch.randn(n, 3, h, w, generator=g).cuda()
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).half().cuda()
    b = torch.randn(64, generator=g).half().cuda()
    got = stem_conv_pool(img, wt, b)
    conv = F.conv2d(img.half().float(), wt.float(), None, stride=2, padding=3)
    want = F.max_pool2d((conv.half().float() + b.float()[None, :, None, None]).relu(), 3, 2, 1)
    assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float() - want).abs()
    assert float((err - 2e-3 * want.abs()).max()) <= 2e-3
    x16 = F.conv2d(img.half().contiguous(memory_format=torch.channels_last), wt, None, stride=2, padding=3)
    if x16.is_contiguous(memory_format=torch.channels_last):
        old = bias_relu_maxpool(x16, b)
        assert float(((got.float() - old.float()).abs() - 2e-3 * old.float().abs()).max()) <= 2e-3
    # a non-contiguous image (a camera slice of a batch) goes through the strides
    wide = torch.randn(n, 5, h, w + 3, generator=g).cuda()
    view = wide[:, 1:4, :, 2:w + 2]
    assert torch.equal(stem_conv_pool(view, wt, b), stem_conv_pool(view.contiguous(), wt, b))


@gpu
def test_warm_frame_reaches_no_vendor_convolution_or_gemm(monkeypatch):
    """Every matrix / convolution kernel of a frame is this repository's: with F.conv2d, F.linear, torch.matmul / bmm /
    addmm and scaled_dot_product_attention patched to raise, image -> detections still runs (backbone + FPN + decoder +
    decode at R50 704x256, two warm frames after the set-up frames that build the folded weights). The rule behind it: a vendor solver may issue the gfx950 double-K
    matrix instructions, which corrupt other kernels running beside them, and MIOpen picks solvers
    at run time."""
    from simpb_amd import configs, plugin
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    model = model.cuda().fuse_conv_bn().half_backbone()
    imgs = [synth.images(1, f, (704, 256)).cuda() for f in range(4)]
    from tests.helpers import metas_to
    metas = [metas_to(synth.frame_metas(1, f, (704, 256)), "cuda") for f in range(4)]
    with torch.no_grad():   # set-up outside the rule: the first calls build the folded / packed weights (float64 matmuls, once)
        for f in range(2):
            model.simple_test(imgs[f], **metas[f])

    def forbid(name):
        def raiser(*a, **k):
            raise AssertionError(f"{name} reached inside a frame")
        return raiser
    for mod, name in ((F, "conv2d"), (F, "linear"), (F, "scaled_dot_product_attention"), (torch, "matmul"), (torch, "bmm"),
                      (torch, "addmm"), (torch, "mm"), (torch, "baddbmm"), (torch.Tensor, "matmul"), (torch.Tensor, "__matmul__")):
        monkeypatch.setattr(mod, name, forbid(f"{mod.__name__}.{name}"))
    with torch.no_grad():
        for f in range(2, 4):
            res = model.simple_test(imgs[f], **metas[f])
            assert res[0]["img_bbox"]["boxes_3d"].shape == (300, 10)


@gpu
def test_conv3x3_writes_tokens():
    """FPN.fpn_convs writing the decoder's token buffer themselves: each level's convolution with `tokens` must leave in
    col_feats exactly what feature_maps_format (ops/__init__.py:63-92) makes of the f16 maps the same kernel writes."""
    from simpb_amd.plugin.ops import conv3x3_nhwc, feature_maps_format
    g = torch.Generator().manual_seed(5)
    bs, cams, c = 2, 3, 256
    shapes = [(16, 44), (8, 22), (4, 11), (2, 6)]
    per_cam = sum(h * w for h, w in shapes)
    col = torch.full((bs, cams * per_cam, c), float("nan"), device="cuda")
    col16 = torch.full((bs, cams * per_cam, c), float("nan"), device="cuda", dtype=torch.float16)
    maps, start = [], 0
    for h, w in shapes:
        x = torch.randn(bs * cams, c, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(c, c, 3, 3, generator=g) / 48).half().cuda().contiguous(memory_format=torch.channels_last)
        b = torch.randn(c, generator=g).half().cuda()
        maps.append(conv3x3_nhwc(x, wt, b, relu=False))
        assert conv3x3_nhwc(x, wt, b, relu=False, tokens=(col, per_cam, start, col16)) is None
        start += h * w
    want = feature_maps_format([m.float().reshape(bs, cams, c, *m.shape[2:]) for m in maps])[0]
    assert torch.equal(col, want)
    assert torch.equal(col16.float(), want)      # the same rows without the widening (value_proj's input)
    with pytest.raises(ValueError):
        conv3x3_nhwc(x, wt, b, relu=False, tokens=(col, per_cam, per_cam - 1))   # level does not fit its block


@gpu
@pytest.mark.parametrize("shapes,images", [([(64, 176), (32, 88), (16, 44), (8, 22)], 6), ([(16, 44), (8, 22), (4, 11), (2, 6)], 12),
                                           ([(5, 7), (3, 3)], 2)])
def test_conv3x3_group_writes_the_tokens_of_every_level_in_one_launch(shapes, images):
    """The FPN's four output convolutions as ONE launch (simpb_conv3x3_group_tokens_f16): the same 96 x 128 staged tiles as
    tiling 8 of the single call, so every level's rows must EQUAL those the per-level calls with variant 8 write, f32 and
    f16, at the shipped R50 map sizes, at small ones and with ragged tiles; rows of no level stay untouched."""
    from simpb_amd.plugin.ops import conv3x3_group_tokens, conv3x3_nhwc
    g = torch.Generator().manual_seed(len(shapes) * 7 + images)
    c = 256
    per_cam = sum(h * w for h, w in shapes) + 3          # (three spare rows per camera: nobody's)
    starts = [sum(h * w for h, w in shapes[:i]) for i in range(len(shapes))]
    xs = [torch.randn(images, c, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last) for h, w in shapes]
    ws = [(torch.randn(c, c, 3, 3, generator=g) / 48).half().cuda().contiguous(memory_format=torch.channels_last) for _ in shapes]
    bs_ = [torch.randn(c, generator=g).half().cuda() for _ in shapes]
    want = torch.full((1, images * per_cam, c), -7.0, device="cuda")
    want16 = torch.full((1, images * per_cam, c), -7.0, device="cuda", dtype=torch.float16)
    for x, w, b, st in zip(xs, ws, bs_, starts):
        conv3x3_nhwc(x, w, b, relu=False, tokens=(want, per_cam, st, want16), variant=8)
    got, got16 = torch.full_like(want, -7.0), torch.full_like(want16, -7.0)
    conv3x3_group_tokens(xs, ws, bs_, got, per_cam, starts, got16)
    assert torch.equal(got, want) and torch.equal(got16, want16)
    assert float(got.view(images, per_cam, c)[:, per_cam - 3:].max()) == -7.0
    ref = F.conv2d(xs[-1].float(), ws[-1].float(), bs_[-1].float(), padding=1)          # and one level against fp32
    h, w = shapes[-1]
    rows = got.view(images, per_cam, c)[:, starts[-1]: starts[-1] + h * w].reshape(images, h, w, c).permute(0, 3, 1, 2)
    assert float((rows - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))


@gpu
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("cin,cout,h,w", [(512, 256, 32, 88), (256, 256, 64, 176), (1024, 256, 16, 44), (192, 40, 6, 10)])
def test_conv1x1_with_upsampled_residual(cin, cout, h, w, variant):
    """The FPN top-down sum lateral[i-1] = conv1x1(c[i-1]) + interpolate(lateral[i], nearest) (mmdet FPN.forward) with
    the 2x nearest read done inside the launch, against the materialised F.interpolate + conv1x1 residual route (same
    kernel, same arithmetic: bit-exact) and against fp32 F.conv2d."""
    from simpb_amd.plugin.ops import conv1x1_nhwc
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(3, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().cuda()
    b = torch.randn(cout, generator=g).half().cuda()
    coarse = torch.randn(3, cout, h // 2, w // 2, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    up = F.interpolate(coarse, size=(h, w), mode="nearest").contiguous(memory_format=torch.channels_last)
    want = conv1x1_nhwc(x, wt, b, up, relu=False, variant=variant)
    got = conv1x1_nhwc(x, wt, b, coarse, relu=False, residual_upsample2x=True, variant=variant)
    assert torch.equal(got, want)
    ref = F.conv2d(x.float(), wt.float(), b.float()) + up.float()
    assert float((got.float() - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        conv1x1_nhwc(x, wt, b, up, relu=False, residual_upsample2x=True)   # residual must be the half-size map


@gpu
def test_format_tokens_with_level_bias_equals_bias_then_format():
    """The FPN's output convolutions run without their bias and ops.format_tokens adds it while it writes the tokens:
    same numbers as conv + fp16 bias add + format (the sum is rounded to fp16 first), for fp16 and fp32 levels; and the
    whole fused neck equals the unfused neck on the same fp16 backbone features."""
    from simpb_amd.plugin import ops
    shapes = [(8, 22), (4, 11), (2, 6), (1, 3)]
    for dtype in (torch.float16, torch.float32):
        maps = [torch.from_numpy(synth.randn(f"fmtb.l{l}", (12, 16, h, w))).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
                for l, (h, w) in enumerate(shapes)]
        bias = [torch.from_numpy(synth.randn(f"fmtb.b{l}", (16,))).to(dtype).cuda() for l in range(4)]
        want = ops.format_tokens([(m + b.view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last) for m, b in zip(maps, bias)], 2, 6)
        got = ops.format_tokens(maps, 2, 6, biases=bias)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        part = ops.format_tokens(maps, 2, 6, biases=[bias[0], None, None, bias[3]])[0]
        mixed = ops.format_tokens([(maps[0] + bias[0].view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last), maps[1], maps[2],
                                   (maps[3] + bias[3].view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last)], 2, 6)[0]
        assert torch.equal(part, mixed)


@gpu
def test_fused_neck_equals_unfused_neck():
    from simpb_amd import configs, plugin
    from simpb_amd.plugin import detector
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    model = model.cuda().fuse_conv_bn().half_backbone()
    img = synth.images(1, 1, (352, 128)).cuda()
    with torch.no_grad():
        got = model.extract_feat(img)           # conv1x1 + conv3x3 kernels, the FPN writes the tokens itself
        assert not model.img_neck.deferred_output_bias
        from simpb_amd.plugin import routes
        with routes.override(conv3x3_kernel=False, stem_kernel=False):   # vendor 3x3 convolutions, biases added by the token format pass
            mid = model.extract_feat(img)
            assert model.img_neck.deferred_output_bias
            with routes.override(conv1x1_kernel=False):   # mmdet's statement: lateral convs, F.interpolate, adds, biased 3x3 convs
                want = model.extract_feat(img)
        assert not model.img_neck.deferred_output_bias
    scale = float(want[0].abs().max())
    assert float((mid[0] - want[0]).abs().max()) <= 2e-2 * scale
    assert float((mid[0] - want[0]).abs().mean()) <= 2e-3 * scale
    assert torch.equal(mid[1], want[1]) and torch.equal(mid[2], want[2])
    scale = float(want[0].abs().max())
    assert float((got[0] - want[0]).abs().max()) <= 2e-2 * scale      # fp16 networks, different rounding points
    assert float((got[0] - want[0]).abs().mean()) <= 2e-3 * scale
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])


@gpu
@pytest.mark.parametrize("cin,cout,h,w,res", [(64, 256, 64, 176, True), (512, 2048, 8, 22, True), (128, 512, 32, 88, False), (192, 40, 5, 7, True)])
def test_conv1x1_with_input_bias_relu(cin, cout, h, w, res):
    """conv3 of a bottleneck applying conv2's folded-BN bias + ReLU while it stages its input, against the separate
    bias_act pass followed by the same kernel (fp32 add, one rounding to fp16 on both routes: bit-exact)."""
    from simpb_amd.plugin.ops import bias_act_, conv1x1_nhwc
    g = torch.Generator().manual_seed(cin * 3 + h)
    x = torch.randn(3, cin, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last)
    b_in = torch.randn(cin, generator=g).half().cuda()
    wt = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).half().cuda()
    b = torch.randn(cout, generator=g).half().cuda()
    r = torch.randn(3, cout, h, w, generator=g).half().cuda().contiguous(memory_format=torch.channels_last) if res else None
    want = conv1x1_nhwc(bias_act_(x.clone(), b_in, None, relu=True), wt, b, r, relu=True)
    got = conv1x1_nhwc(x, wt, b, r, relu=True, input_bias=b_in)
    assert torch.equal(got, want)


@gpu
def test_vendor_convolution_fallback_warns_and_is_refused_beside_a_decoder():
    """A bottleneck whose channel counts miss the in-tree kernels' rules (48 input channels) goes to the vendor convolution:
    with a RuntimeWarning when the module runs on its own, with a RuntimeError once a pipelined runner has declared that the
    backbone runs beside a decoder (plugin/detector.py STRICT_NO_VENDOR), and silently when the route was switched off on
    purpose (the tests' vendor cross-check)."""
    import warnings
    from simpb_amd.plugin import detector, routes
    blk = detector.Bottleneck(48, 16).cuda().half().eval().to(memory_format=torch.channels_last)
    for m in (blk.conv1, blk.conv2, blk.conv3):   # as SimPB.fuse_conv_bn leaves them: biases instead of BatchNorm
        m.bias = torch.nn.Parameter(torch.zeros(m.out_channels, device="cuda", dtype=torch.float16))
    blk.bn1 = blk.bn2 = blk.bn3 = torch.nn.Identity()
    blk.downsample = torch.nn.Sequential(torch.nn.Conv2d(48, 64, 1, bias=True).cuda().half())
    blk.fused_epilogue = True
    x = torch.randn(2, 48, 8, 8, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    old, old_seen = detector.STRICT_NO_VENDOR, set(detector._warned)
    try:
        detector.STRICT_NO_VENDOR = False
        detector._warned.clear()
        with torch.no_grad(), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            y = blk(x)
        assert y.shape == (2, 64, 8, 8) and any("vendor convolution" in str(w.message) for w in caught)
        detector.STRICT_NO_VENDOR = True
        with torch.no_grad(), pytest.raises(RuntimeError, match="pipelined runner"):
            blk(x)
        with torch.no_grad(), routes.override(conv1x1_kernel=False, conv3x3_kernel=False), warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            blk(x)
        assert not any("vendor convolution" in str(w.message) for w in caught)
    finally:
        detector.STRICT_NO_VENDOR = old
        detector._warned.clear()
        detector._warned.update(old_seen)
