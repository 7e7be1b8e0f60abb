"""Operator layer: same names and argument meaning as the reference's
projects/mmdet3d_plugin/ops/__init__.py and ops/deformable_aggregation.py, bound to the C-ABI HIP
library (include/simpb_hip.h) instead of the pybind11 CUDA extension. No CPU path exists here:
tensors must live on the GPU and the library must be built."""
import ctypes

import torch
from torch.autograd.function import Function

from .. import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("simpb_amd operators run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback")


_layout_ok = {}


def _check_layout(spatial_shape, scale_start_index, num_feat):
    """Host-side bounds check of the (H, W, start) table the kernel will index with. Costs one
    device->host copy the first time a given table tensor is seen; the head reuses its tables."""
    key = (spatial_shape.data_ptr(), scale_start_index.data_ptr(), spatial_shape._version,
           scale_start_index._version, num_feat)
    if key in _layout_ok:
        return
    ss = spatial_shape.detach().cpu().long()
    st = scale_start_index.detach().cpu().long()
    if bool((ss <= 0).any()) or bool((st < 0).any()) or int((st + ss[..., 0] * ss[..., 1]).max()) > num_feat:
        raise ValueError("spatial_shape/scale_start_index address tokens outside mc_ms_feat")
    if len(_layout_ok) > 64:
        _layout_ok.clear()
    _layout_ok[key] = True


class DeformableAggregationFunction(Function):
    """ops/deformable_aggregation.py:7-75: forward and backward of the operator."""

    @staticmethod
    def forward(ctx, mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
        _require_gpu(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights)
        mc_ms_feat = mc_ms_feat.contiguous().float()
        spatial_shape = spatial_shape.contiguous().int()
        scale_start_index = scale_start_index.contiguous().int()
        sampling_location = sampling_location.contiguous().float()
        weights = weights.contiguous().float()
        bs, num_feat, num_embeds = mc_ms_feat.shape
        num_cams, num_scale = spatial_shape.shape[:2]
        _, num_anchors, num_pts = sampling_location.shape[:3]
        num_groups = weights.shape[5]
        # the kernel indexes with these shapes; refuse anything inconsistent before launching
        if tuple(spatial_shape.shape) != (num_cams, num_scale, 2) or tuple(scale_start_index.shape) != (num_cams, num_scale):
            raise ValueError("spatial_shape must be [cam, lvl, 2] and scale_start_index [cam, lvl]")
        if tuple(sampling_location.shape) != (bs, num_anchors, num_pts, num_cams, 2):
            raise ValueError(f"sampling_location must be [bs, anchor, pts, cam, 2], got {tuple(sampling_location.shape)}")
        if tuple(weights.shape) != (bs, num_anchors, num_pts, num_cams, num_scale, num_groups):
            raise ValueError(f"weights must be [bs, anchor, pts, cam, lvl, group], got {tuple(weights.shape)}")
        if num_embeds % num_groups != 0:
            raise ValueError("num_embeds must be divisible by num_groups")
        _check_layout(spatial_shape, scale_start_index, num_feat)
        output = torch.empty(bs, num_anchors, num_embeds, device=mc_ms_feat.device, dtype=torch.float32)
        status = _lib.lib().simpb_deformable_aggregation_forward(
            _ptr(output), _ptr(mc_ms_feat), _ptr(spatial_shape), _ptr(scale_start_index), _ptr(sampling_location),
            _ptr(weights), bs, num_cams, num_feat, num_embeds, num_scale, num_anchors, num_pts, num_groups, _stream())
        _lib.check(status, "simpb_deformable_aggregation_forward")
        if any(t.requires_grad for t in (mc_ms_feat, sampling_location, weights)):
            ctx.save_for_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        """ops/deformable_aggregation.py:39-75."""
        mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights = ctx.saved_tensors
        grad_output = grad_output.contiguous().float()
        bs, num_feat, num_embeds = mc_ms_feat.shape
        num_cams, num_scale = spatial_shape.shape[:2]
        _, num_anchors, num_pts = sampling_location.shape[:3]
        num_groups = weights.shape[5]
        grad_feat = torch.empty_like(mc_ms_feat)
        grad_loc = torch.empty_like(sampling_location)
        grad_w = torch.empty_like(weights)
        status = _lib.lib().simpb_deformable_aggregation_backward(
            _ptr(grad_feat), _ptr(grad_loc), _ptr(grad_w), _ptr(mc_ms_feat), _ptr(spatial_shape), _ptr(scale_start_index),
            _ptr(sampling_location), _ptr(weights), _ptr(grad_output), bs, num_cams, num_feat, num_embeds, num_scale,
            num_anchors, num_pts, num_groups, _stream())
        _lib.check(status, "simpb_deformable_aggregation_backward")
        return grad_feat, None, None, grad_loc, grad_w


def deformable_aggregation_function(feature_maps, spatial_shape, scale_start_index, sampling_location, weights):
    """ops/__init__.py:6-19."""
    return DeformableAggregationFunction.apply(feature_maps, spatial_shape, scale_start_index, sampling_location, weights)


def dfa_fused(feat, spatial_shape, scale_start_index, anchor, learn, fix_scale, proj, image_wh, feat_logits, cam_logits,
              num_groups, want_operands=False):
    """DeformableFeatureAggregation between its Linear layers as ONE launch (csrc/deform_agg_fused.hip): key points,
    projection, weight softmax and the aggregation. feat: the token buffer, f32 or f16 [bs, num_feat, C]. Returns the
    aggregated features [bs, A, C] (and, with want_operands, the sampling locations and weights the launch used, in the
    drop-in operator's layouts)."""
    _require_gpu(feat, anchor, learn, feat_logits, cam_logits, proj, image_wh)
    if feat.dtype not in (torch.float32, torch.float16) or not feat.is_contiguous():
        raise ValueError("dfa_fused: contiguous f32 or f16 token buffer expected")
    bs, num_feat, c = feat.shape
    cams, lvls = spatial_shape.shape[:2]
    a = anchor.shape[1]
    num_fix, num_learn = fix_scale.shape[0], learn.shape[-1] // 3
    p = num_fix + num_learn
    spatial_shape = spatial_shape.contiguous().int()
    scale_start_index = scale_start_index.contiguous().int()
    anchor, learn = anchor.contiguous().float(), learn.contiguous().float()
    feat_logits, cam_logits = feat_logits.contiguous().float(), cam_logits.contiguous().float()
    proj, image_wh = proj.contiguous().float(), image_wh.contiguous().float()
    fix_scale = fix_scale.contiguous().float()
    lpg = lvls * p * num_groups
    if (tuple(anchor.shape) != (bs, a, 11) or tuple(learn.shape) != (bs, a, num_learn * 3) or tuple(feat_logits.shape) != (bs, a, lpg)
            or tuple(cam_logits.shape) != (bs, cams, lpg) or tuple(proj.shape) != (bs, cams, 4, 4) or tuple(image_wh.shape) != (bs, cams, 2)
            or tuple(spatial_shape.shape) != (cams, lvls, 2) or tuple(scale_start_index.shape) != (cams, lvls)):
        raise ValueError("dfa_fused: operand shapes disagree")
    _check_layout(spatial_shape, scale_start_index, num_feat)
    out = torch.empty(bs, a, c, device=feat.device, dtype=torch.float32)
    loc = torch.empty(bs, a, p, cams, 2, device=feat.device) if want_operands else None
    w = torch.empty(bs, a, p, cams, lvls, num_groups, device=feat.device) if want_operands else None
    status = _lib.lib().simpb_dfa_fused_forward(
        _ptr(out), _ptr(feat), 1 if feat.dtype == torch.float16 else 0, _ptr(spatial_shape), _ptr(scale_start_index),
        _ptr(anchor), _ptr(learn), _ptr(fix_scale), _ptr(proj), _ptr(image_wh), _ptr(feat_logits), _ptr(cam_logits),
        _ptr(loc) if loc is not None else None, _ptr(w) if w is not None else None, bs, cams, num_feat, c, lvls, a, num_fix,
        num_learn, num_groups, _stream())
    _lib.check(status, "simpb_dfa_fused_forward")
    return (out, loc, w) if want_operands else out


def dfa_locations(anchor, learn, fix_scale, proj, image_wh):
    """Sampling locations [bs, A, P, cams, 2] of DeformableFeatureAggregation's key points (csrc/dfa_prep.hip:
    dfa_points): what the one-launch form computes on chip; used by measurement code to count valid samples."""
    _require_gpu(anchor, learn, proj, image_wh)
    anchor, learn = anchor.contiguous().float(), learn.contiguous().float()
    proj, image_wh, fix_scale = proj.contiguous().float(), image_wh.contiguous().float(), fix_scale.contiguous().float()
    bs, a = anchor.shape[:2]
    cams = proj.shape[1]
    num_fix, num_learn = fix_scale.shape[0], learn.shape[-1] // 3
    loc = torch.empty(bs, a, num_fix + num_learn, cams, 2, device=anchor.device)
    _lib.check(_lib.lib().simpb_dfa_points(_ptr(loc), None, _ptr(anchor), _ptr(learn), _ptr(fix_scale), _ptr(proj), _ptr(image_wh),
                                           bs, a, num_fix, num_learn, cams, _stream()), "simpb_dfa_points")
    return loc


def feature_maps_format(feature_maps, inverse=False):
    """ops/__init__.py:22-92. Forward direction: list of [bs, cam, C, H, W] ->
    [col_feats [bs, sum(cam*H*W), C], spatial_shape i64[cam, lvl, 2], scale_start_index i64[cam, lvl]].
    Maps that are already channels_last in memory make the permute a cheap strided copy."""
    if inverse:
        col_feats, spatial_shape, scale_start_index = feature_maps
        num_cams, num_levels = spatial_shape.shape[:2]
        shapes = spatial_shape[0].tolist()
        if not bool((spatial_shape == spatial_shape[:1]).all()):
            raise NotImplementedError("inverse format with per-camera shapes")
        bs = col_feats.shape[0]
        per_cam = col_feats.reshape(bs, num_cams, -1, col_feats.shape[-1])
        out, start = [], 0
        for h, w in shapes:
            out.append(per_cam[:, :, start:start + h * w].reshape(bs, num_cams, h, w, -1).permute(0, 1, 4, 2, 3))
            start += h * w
        return [out]

    if isinstance(feature_maps[0], (list, tuple)):
        formated = [feature_maps_format(x) for x in feature_maps]
        return [torch.cat([x[0] for x in formated], dim=1), torch.cat([x[1] for x in formated], dim=0),
                torch.cat([x[2] for x in formated], dim=0)]

    bs, num_cams = feature_maps[0].shape[:2]
    shapes = tuple(tuple(f.shape[-2:]) for f in feature_maps)
    col = torch.cat([f.reshape(bs, num_cams, f.shape[2], -1) for f in feature_maps], dim=-1)
    col = col.permute(0, 1, 3, 2).flatten(1, 2)
    spatial_shape, scale_start_index = _shape_tables(shapes, num_cams, col.device)
    return [col, spatial_shape, scale_start_index]


_table_cache = {}


def _shape_tables(shapes, num_cams, device):
    """(spatial_shape i64[cam, lvl, 2], scale_start_index i64[cam, lvl]) for a pyramid, built once
    per (shapes, cams, device): the tables are constants of the configuration, and building them
    per frame would put a host->device copy inside every frame (and inside a graph capture)."""
    key = (shapes, num_cams, str(device))
    if key not in _table_cache:
        spatial_shape = torch.tensor([list(map(list, shapes))] * num_cams, dtype=torch.int64)
        sizes = [h * w for h, w in shapes] * num_cams
        starts = [0]
        for s in sizes[:-1]:
            starts.append(starts[-1] + s)
        start = torch.tensor(starts, dtype=torch.int64).reshape(num_cams, -1)
        _table_cache[key] = (spatial_shape.to(device), start.to(device))
    return _table_cache[key]


def token_tables(shapes, num_cams, device):
    """(spatial_shape, scale_start_index) of feature_maps_format (ops/__init__.py:80-90) for level shapes ((h, w), ...)."""
    return _shape_tables(tuple(tuple(s) for s in shapes), num_cams, device)


def query_cam_from_groups(query_groups, num_query, device):
    """[(start, end)] * num_cams (allocation.py:99) -> i32[num_query] camera id per query slot."""
    cam = torch.zeros(num_query, dtype=torch.int32)
    for i, (s, e) in enumerate(query_groups):
        cam[s:e] = i
    return cam.to(device)


class MSDeformAttnGroupedFunction(Function):
    """Grouped counterpart of mmcv's MultiScaleDeformableAttnFunction (forward + backward)."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, sampling_locations, attention_weights, query_cam):
        bs, num_cams, num_value, heads, ch = value.shape
        _, nq, _, lvls, pts, _ = sampling_locations.shape
        output = torch.empty(bs, nq, heads * ch, device=value.device, dtype=torch.float32)
        if nq:
            status = _lib.lib().simpb_ms_deform_attn_grouped_forward(
                _ptr(output), _ptr(value), _ptr(spatial_shapes), _ptr(level_start_index), _ptr(sampling_locations),
                _ptr(attention_weights), _ptr(query_cam), bs, num_cams, num_value, heads, ch, lvls, pts, nq, _stream())
            _lib.check(status, "simpb_ms_deform_attn_grouped_forward")
        if any(t.requires_grad for t in (value, sampling_locations, attention_weights)):
            ctx.save_for_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, query_cam)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights, query_cam = ctx.saved_tensors
        bs, num_cams, num_value, heads, ch = value.shape
        _, nq, _, lvls, pts, _ = sampling_locations.shape
        g_value = torch.empty_like(value)
        g_loc = torch.empty_like(sampling_locations)
        g_attn = torch.empty_like(attention_weights)
        status = _lib.lib().simpb_ms_deform_attn_grouped_backward(
            _ptr(g_value), _ptr(g_loc), _ptr(g_attn), _ptr(value), _ptr(spatial_shapes), _ptr(level_start_index),
            _ptr(sampling_locations), _ptr(attention_weights), _ptr(query_cam), _ptr(grad_output.contiguous().float()),
            bs, num_cams, num_value, heads, ch, lvls, pts, nq, _stream())
        _lib.check(status, "simpb_ms_deform_attn_grouped_backward")
        return g_value, None, None, g_loc, g_attn, None


def ms_deform_attn_grouped(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, query_cam):
    """One launch for what group_attn.py:227-235 does with a Python loop over cameras and
    MultiScaleDeformableAttnFunction.apply. value [bs, cam, Nv, heads, ch]; sampling_locations
    [bs, Nq, heads, lvl, pts, 2]; attention_weights [bs, Nq, heads, lvl, pts]; query_cam i32[Nq]
    -> [bs, Nq, heads*ch]."""
    _require_gpu(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, query_cam)
    value = value.contiguous().float()
    spatial_shapes = spatial_shapes.contiguous().long()
    level_start_index = level_start_index.contiguous().long()
    sampling_locations = sampling_locations.contiguous().float()
    attention_weights = attention_weights.contiguous().float()
    query_cam = query_cam.contiguous().int()
    bs, num_cams, num_value, heads, ch = value.shape
    _, nq, _, lvls, pts, _ = sampling_locations.shape
    if tuple(sampling_locations.shape) != (bs, nq, heads, lvls, pts, 2):
        raise ValueError(f"sampling_locations must be [bs, Nq, heads, lvl, pts, 2], got {tuple(sampling_locations.shape)}")
    if tuple(attention_weights.shape) != (bs, nq, heads, lvls, pts):
        raise ValueError(f"attention_weights must be [bs, Nq, heads, lvl, pts], got {tuple(attention_weights.shape)}")
    if tuple(spatial_shapes.shape) != (lvls, 2) or tuple(level_start_index.shape) != (lvls,):
        raise ValueError("spatial_shapes must be [lvl, 2] and level_start_index [lvl]")
    if query_cam.numel() != nq:
        raise ValueError("query_cam must have one entry per query slot")
    if ch % 4 != 0:
        raise ValueError("channels per head must be a multiple of 4")
    _check_layout(spatial_shapes[None], level_start_index[None], num_value)
    return MSDeformAttnGroupedFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                             attention_weights, query_cam)


MSDA_LINEAR_WIDTH = 8 * 256 + 128   # row of simpb_msda_linear_forward: 8 heads x 256 channel sums | 8 tap-weight sums | pad (a 128-deep chunk)


def msda_linear(tokens, spatial_shapes, level_start_index, raw, reference_points, query_cam, m_live=None):
    """Camera-grouped deformable sampling of the RAW camera tokens (csrc/msda_lin.hip): what
    QueryGroupMultiScaleDeformableAttention computes between its value_proj and its output_proj, with value_proj moved
    behind the sampling by linearity. tokens f16 or f32 [bs, cams, Nv, 256]; raw [bs, Nq, 384] = sampling_offsets |
    attention logits of [query | pos]; reference_points [bs, Nq, (1,) 2]; -> agg f32 [bs, Nq, 2176] (with m_live the
    rows of capacity slots are left unwritten: the product behind it skips them; without it they are zeros)."""
    _require_gpu(tokens, raw, reference_points, query_cam)
    if tokens.dtype not in (torch.float16, torch.float32) or not tokens.is_contiguous() or tokens.dim() != 4 or tokens.shape[-1] != 256:
        raise ValueError("msda_linear: contiguous f16 / f32 tokens [bs, cams, Nv, 256] expected")
    bs, cams, nv, _ = tokens.shape
    nq = raw.shape[1]
    spatial_shapes = spatial_shapes.contiguous().long()
    level_start_index = level_start_index.contiguous().long()
    if tuple(spatial_shapes.shape) != (4, 2) or tuple(level_start_index.shape) != (4,) or raw.shape[0] != bs or raw.shape[-1] != 384:
        raise ValueError("msda_linear: the shipped layout (4 levels, 8 heads x 4 levels x 4 points) expected")
    _check_layout(spatial_shapes[None], level_start_index[None], nv)
    from .dense import rows2d
    rawt, rows, ldraw = rows2d(raw)
    reft, rrows, ldref = rows2d(reference_points.reshape(bs, nq, -1)[..., :2])
    query_cam = query_cam.contiguous().int()
    if rows != bs * nq or rrows != bs * nq or query_cam.numel() != nq:
        raise ValueError("msda_linear: one raw row, one reference point and one camera per query slot")
    # the kernel leaves the rows of capacity slots (query_cam < 0, slots behind m_live) unwritten. With m_live the product
    # behind it skips them; without it (the reference's padded batch) that product reads every row, so they must hold
    # numbers: zeros, what the grouped sampler (csrc/msda.hip) writes there
    agg = (torch.empty if m_live is not None else torch.zeros)(bs, nq, MSDA_LINEAR_WIDTH, device=tokens.device, dtype=torch.float32)
    status = _lib.lib().simpb_msda_linear_forward(
        _ptr(agg), MSDA_LINEAR_WIDTH, _ptr(tokens), 1 if tokens.dtype == torch.float16 else 0, _ptr(spatial_shapes),
        _ptr(level_start_index), _ptr(rawt), ldraw, _ptr(reft), ldref, _ptr(query_cam),
        _ptr(m_live) if m_live is not None else None, bs, cams, nv, 8, 32, 4, 4, nq, _stream())
    _lib.check(status, "simpb_msda_linear_forward")
    return agg


def linear_f32(x, weight, bias=None, relu=False):
    """F.linear(x, weight, bias) (optionally + ReLU) in exact fp32 on the f32 matrix cores
    (csrc/linear.hip). x [..., K], weight [N, K]; K must be a multiple of 32."""
    _require_gpu(x, weight)
    k = x.shape[-1]
    n = weight.shape[0]
    if weight.shape[1] != k or k % 32 != 0:
        raise ValueError(f"linear_f32 needs weight [N, K] with K % 32 == 0, got x {tuple(x.shape)} w {tuple(weight.shape)}")
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    x2, weight = x2.float(), weight.contiguous().float()
    if bias is not None:
        bias = bias.contiguous().float()
        if bias.numel() != n:
            raise ValueError("bias must have N elements")
    m = x2.shape[0]
    y = torch.empty(m, n, device=x.device, dtype=torch.float32)
    if m:
        status = _lib.lib().simpb_linear_f32(_ptr(y), _ptr(x2), _ptr(weight), _ptr(bias) if bias is not None else None,
                                             m, n, k, 1 if relu else 0, _stream())
        _lib.check(status, "simpb_linear_f32")
    return y.reshape(x.shape[:-1] + (n,))


def attention_f32(q, k, v, num_heads, query_cam=None, group_start=None, split=None):
    """softmax(q k^T / sqrt(hd)) v per head, fp32 in and out, flash style (csrc/attention.hip). q [bs, Nq, E],
    k/v [bs, Nk, E] with unit inner stride (row-strided views of a fused projection are fine), E =
    num_heads * 64. With query_cam/group_start: camera-grouped self-attention over one slot set.
    split: 0 / False = the exact-fp32 matrix instruction; 1 / True = the FP16 matrix cores with operands split inside the
    kernel (fp32-grade); 2 = q / k / v ARE already the half pairs a GEMM with split_halfs=True left (plugin/dense.py; the
    softmax scale folded into q): what a frame runs under routes.attention_split_fp16. None: 1 if that route is on."""
    _require_gpu(q, k, v)
    bs, nq, e = q.shape
    nk = k.shape[1]
    hd = e // num_heads
    if hd != 64 or e != num_heads * 64 or k.shape[2] != e or v.shape[2] != e or v.shape[1] != nk:
        raise ValueError("attention_f32 needs head_dim 64 and matching q/k/v widths")

    def rows(t, n):
        if t.dtype != torch.float32:
            t = t.float()
        if t.stride(2) != 1 or (bs > 1 and t.stride(0) != n * t.stride(1)) or t.stride(1) % 4 or t.data_ptr() % 16:
            t = t.contiguous()
        return t, t.stride(1)

    q, ldq = rows(q, nq)
    k, ldk = rows(k, nk)
    v, ldv = rows(v, nk)
    out = torch.empty(bs, nq, e, device=q.device, dtype=torch.float32)
    if (query_cam is None) != (group_start is None):
        raise ValueError("query_cam and group_start go together")
    if query_cam is not None:
        if nk != nq or query_cam.dtype != torch.int32 or group_start.dtype != torch.int32 or query_cam.numel() != nq:
            raise ValueError("grouped attention needs i32 query_cam [N] / group_start [cams+1] over one slot set")
    if nq == 0:
        return out
    from . import routes
    mode = (1 if routes.R.attention_split_fp16 else 0) if split is None else int(split)
    tables = (_ptr(query_cam) if query_cam is not None else None, _ptr(group_start) if group_start is not None else None)
    if mode == 2:
        status = _lib.lib().simpb_attention_split_halfs(_ptr(out), _ptr(q), _ptr(k), _ptr(v), *tables, bs, num_heads, hd, nq, nk,
                                                        ldq, ldk, ldv, e, _stream())
    else:
        entry = _lib.lib().simpb_attention_f32_split if mode else _lib.lib().simpb_attention_f32
        status = entry(_ptr(out), _ptr(q), _ptr(k), _ptr(v), *tables, bs, num_heads, hd, nq, nk, ldq, ldk, ldv, e,
                       1.0 / (hd ** 0.5), _stream())
    _lib.check(status, "simpb_attention_f32")
    return out


def format_tokens(levels, bs, num_cams, biases=None):
    """feature_maps_format for channels_last level tensors [bs*cams, C, H, W] (f16 or f32), one
    pass (csrc/format.hip). Returns [col_feats, spatial_shape, scale_start_index] like
    feature_maps_format. biases: per-level [C] tensors (same dtype) added on the way, or None."""
    _require_gpu(*levels)
    c = levels[0].shape[1]
    dtype = levels[0].dtype
    shapes = tuple(tuple(f.shape[-2:]) for f in levels)
    srcs = []
    for f in levels:
        if f.shape[0] != bs * num_cams or f.shape[1] != c or f.dtype != dtype:
            raise ValueError("levels must share batch, channels and dtype")
        if not f.is_contiguous(memory_format=torch.channels_last):
            f = f.contiguous(memory_format=torch.channels_last)
        srcs.append(f)
    if dtype not in (torch.float16, torch.float32) or c % 8:
        raise ValueError("format_tokens takes f16/f32 levels with channels % 8 == 0")
    tokens = sum(h * w for h, w in shapes)
    col = torch.empty(bs, num_cams * tokens, c, device=levels[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * len(srcs))(*[f.data_ptr() for f in srcs])
    hws = (ctypes.c_int * len(srcs))(*[h * w for h, w in shapes])
    bptrs = None
    if biases is not None:
        if len(biases) != len(srcs) or any(b is not None and (b.dtype != dtype or b.numel() != c or not b.is_contiguous()) for b in biases):
            raise ValueError("biases: one contiguous [C] tensor of the levels' dtype per level (or None)")
        bptrs = (ctypes.c_void_p * len(srcs))(*[b.data_ptr() if b is not None else None for b in biases])
    status = _lib.lib().simpb_format_tokens(_ptr(col), ptrs, bptrs, hws, len(srcs), bs * num_cams, c,
                                            1 if dtype == torch.float16 else 0, _stream())
    _lib.check(status, "simpb_format_tokens")
    spatial_shape, scale_start_index = _shape_tables(shapes, num_cams, col.device)
    return [col, spatial_shape, scale_start_index]


def bias_act_(y, bias, residual=None, relu=True):
    """In place y = relu?(y + bias[c] + residual?) for a channels_last f16 tensor [N, C, H, W]
    (csrc/bias_act.hip): the epilogue of a BN-folded convolution in one pass."""
    _require_gpu(y, bias)
    n, c, h, w = y.shape
    if (y.dtype != torch.float16 or not y.is_contiguous(memory_format=torch.channels_last) or bias.dtype != torch.float16
            or bias.numel() != c or c % 8):
        raise ValueError("bias_act_ takes a channels_last f16 tensor and an f16 bias with channels % 8 == 0")
    if residual is not None and (residual.shape != y.shape or residual.dtype != torch.float16
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("residual must match y (channels_last f16)")
    status = _lib.lib().simpb_bias_act_nhwc_f16(_ptr(y), _ptr(bias), _ptr(residual) if residual is not None else None,
                                                n * h * w, c, 1 if relu else 0, _stream())
    _lib.check(status, "simpb_bias_act_nhwc_f16")
    return y


def bias_relu_maxpool(x, bias):
    """maxpool3x3/2(relu(x + bias[c])) for a channels_last f16 tensor [N, C, H, W] in one pass (csrc/bias_act.hip): the
    stem epilogue of the BN-folded ResNet (conv1 -> bn1 -> relu -> maxpool)."""
    _require_gpu(x, bias)
    n, c, h, w = x.shape
    if (x.dtype != torch.float16 or not x.is_contiguous(memory_format=torch.channels_last) or bias.dtype != torch.float16
            or bias.numel() != c or c % 8 or not bias.is_contiguous()):
        raise ValueError("bias_relu_maxpool takes a channels_last f16 tensor and an f16 bias with channels % 8 == 0")
    y = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), device=x.device, dtype=torch.float16,
                    memory_format=torch.channels_last)
    _lib.check(_lib.lib().simpb_bias_relu_maxpool_nhwc_f16(_ptr(y), _ptr(x), _ptr(bias), n, h, w, c, _stream()),
               "simpb_bias_relu_maxpool_nhwc_f16")
    return y


def conv1x1_nhwc(x, weight, bias, residual=None, relu=True, stride=1, residual_upsample2x=False, input_bias=None, variant=0):
    """relu?(conv1x1(x, weight, stride) + bias + residual?) for a channels_last f16 tensor, one launch
    (csrc/conv1x1.hip). x [N, Cin, H, W]; weight [Cout, Cin, 1, 1] f16; bias f16 [Cout]. With residual_upsample2x the
    residual is [N, Cout, H/2, W/2] and is read with nearest-neighbour 2x upsampling (the FPN top-down sum). With
    input_bias (f16 [Cin]) x is first replaced by relu(x + input_bias): the epilogue of the convolution that produced it."""
    _require_gpu(x, weight, bias)
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    if (x.dtype != torch.float16 or not x.is_contiguous(memory_format=torch.channels_last) or weight.dtype != torch.float16
            or bias.dtype != torch.float16 or weight.numel() != cout * cin or cin % 64 or cout % 8 or stride not in (1, 2)):
        raise ValueError("conv1x1_nhwc takes channels_last f16 input, f16 [Cout, Cin, 1, 1] weight, Cin % 64 == 0, Cout % 8 == 0")
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float16, memory_format=torch.channels_last)
    want = y.shape if not residual_upsample2x else (n, cout, ho // 2, wo // 2)
    if residual_upsample2x and (residual is None or ho % 2 or wo % 2):
        raise ValueError("residual_upsample2x needs a residual and even output sizes")
    if residual is not None and (tuple(residual.shape) != tuple(want) or residual.dtype != torch.float16
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("residual must match the output (channels_last f16)")
    if input_bias is not None and (input_bias.dtype != torch.float16 or input_bias.numel() != cin or not input_bias.is_contiguous()):
        raise ValueError("input_bias must be a contiguous f16 [Cin] tensor")
    w2 = weight.reshape(cout, cin)
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    status = _lib.lib().simpb_conv1x1_nhwc_f16(_ptr(y), _ptr(x), _ptr(w2), _ptr(bias), _ptr(residual) if residual is not None else None,
                                               n, h, w, cin, cout, stride, 1 if relu else 0,
                                               1 if residual_upsample2x else 0,
                                               _ptr(input_bias) if input_bias is not None else None, variant, _stream())
    _lib.check(status, "simpb_conv1x1_nhwc_f16")
    return y


def conv3x3_nhwc(x, weight, bias, relu=True, stride=1, tokens=None, variant=0):
    """relu?(conv3x3(x, weight, stride, padding=1) + bias) for a channels_last f16 tensor, one launch (csrc/conv3x3.hip).
    x [N, Cin, H, W]; weight f16 [Cout, Cin, 3, 3] (channels_last, i.e. [Cout][3][3][Cin] in memory); bias f16 [Cout].
    tokens = (col_feats f32 [bs, cams * tokens_per_cam, Cout], tokens_per_cam, level_start[, col_f16]): the result is written
    as fp32 token rows of this level (feature_maps_format layout) instead of a map (and, with col_f16, as the same rows in
    f16), and None is returned."""
    _require_gpu(x, weight, bias)
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    if (x.dtype != torch.float16 or not x.is_contiguous(memory_format=torch.channels_last) or weight.dtype != torch.float16
            or bias.dtype != torch.float16 or tuple(weight.shape) != (cout, cin, 3, 3) or cin % 64 or cout % 8
            or stride not in (1, 2) or bias.numel() != cout or not bias.is_contiguous()):
        raise ValueError("conv3x3_nhwc takes channels_last f16 input, f16 [Cout, Cin, 3, 3] weight, Cin % 64 == 0, Cout % 8 == 0")
    if not weight.is_contiguous(memory_format=torch.channels_last):
        weight = weight.contiguous(memory_format=torch.channels_last)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    col16 = None
    if tokens is None:
        y = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float16, memory_format=torch.channels_last)
        col, per_cam, start = None, 0, 0
    else:
        col, per_cam, start = tokens[:3]
        col16 = tokens[3] if len(tokens) > 3 else None
        y = None
        if col16 is not None and (col16.dtype != torch.float16 or not col16.is_contiguous() or col16.shape != col.shape):
            raise ValueError("tokens: the f16 buffer must match col_feats")
        if (col.dtype != torch.float32 or not col.is_contiguous() or col.shape[-1] != cout or col.numel() != n * per_cam * cout
                or start < 0 or start + ho * wo > per_cam):
            raise ValueError("tokens: (contiguous f32 [bs, cams * tokens_per_cam, Cout] with bs * cams == N, tokens_per_cam, level_start)")
    status = _lib.lib().simpb_conv3x3_nhwc_f16(_ptr(y) if y is not None else None, _ptr(col) if col is not None else None,
                                               _ptr(col16) if col16 is not None else None, per_cam, start, _ptr(x), _ptr(weight), _ptr(bias), n, h, w, cin, cout, stride,
                                               1 if relu else 0, variant, _stream())
    _lib.check(status, "simpb_conv3x3_nhwc_f16")
    return y


def conv3x3_group_tokens(xs, weights, biases, col, per_cam, starts, col16=None, relu=False):
    """conv3x3_nhwc(..., tokens=...) of up to four levels in ONE launch (csrc/conv3x3.hip: conv_staged_group_kernel): the
    FPN's output convolutions. xs: channels_last f16 [N, Cin, H_j, W_j]; weights f16 [Cout, Cin, 3, 3]; biases f16 [Cout];
    col f32 [bs, cams * per_cam, Cout] (and col16, the same rows in f16); starts[j] = level j's first row inside a camera."""
    import ctypes
    _require_gpu(*xs, *weights, *biases, col)
    n, cin = xs[0].shape[:2]
    cout = weights[0].shape[0]
    k = len(xs)
    if not (1 <= k <= 4 and len(weights) == k and len(biases) == k and len(starts) == k):
        raise ValueError("conv3x3_group_tokens: 1..4 levels, one weight / bias / start each")
    ws = []
    for x, w, b in zip(xs, weights, biases):
        if (x.dtype != torch.float16 or not x.is_contiguous(memory_format=torch.channels_last) or x.shape[0] != n or x.shape[1] != cin
                or w.dtype != torch.float16 or tuple(w.shape) != (cout, cin, 3, 3) or b.dtype != torch.float16 or b.numel() != cout
                or not b.is_contiguous() or cin % 64 or cout % 8):
            raise ValueError("conv3x3_group_tokens takes channels_last f16 inputs of one channel count, f16 [Cout, Cin, 3, 3] weights")
        ws.append(w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last))
    if (col.dtype != torch.float32 or not col.is_contiguous() or col.shape[-1] != cout or col.numel() != n * per_cam * cout
            or (col16 is not None and (col16.dtype != torch.float16 or not col16.is_contiguous() or col16.shape != col.shape))):
        raise ValueError("conv3x3_group_tokens: col = contiguous f32 [bs, cams * tokens_per_cam, Cout] with bs * cams == N")
    arr_p, arr_i = ctypes.c_void_p * k, ctypes.c_int * k
    status = _lib.lib().simpb_conv3x3_group_tokens_f16(
        k, _ptr(col), _ptr(col16) if col16 is not None else None, int(per_cam), arr_i(*[int(v) for v in starts]),
        arr_p(*[x.data_ptr() for x in xs]), arr_p(*[w.data_ptr() for w in ws]), arr_p(*[b.data_ptr() for b in biases]), n,
        arr_i(*[x.shape[2] for x in xs]), arr_i(*[x.shape[3] for x in xs]), cin, cout, 1 if relu else 0, _stream())
    _lib.check(status, "simpb_conv3x3_group_tokens_f16")


def _stem_weight_packed(weight):
    """f16 [28][2][64][4] fragment order of csrc/stem.hip from the stem's [64, 3, 7, 7] weight, cached on the tensor."""
    tag = (weight.data_ptr(), weight._version, str(weight.device))
    hit = getattr(weight, "_simpb_stem_pack", None)
    if hit is None or hit[0] != tag:
        with torch.no_grad():
            w = weight.detach().float()                                  # [n, c, ky, kx]
            full = torch.zeros(64, 4, 7, 8, device=w.device)
            full[:, :3, :, :7] = w
            # [ky][g][kb][n][c] with kx = 2 g + kb
            pack = full.reshape(64, 4, 7, 4, 2).permute(2, 3, 4, 0, 1).contiguous().half()
        hit = (tag, pack)
        weight._simpb_stem_pack = hit
    return hit[1]


def stem_conv_pool(img, weight, bias):
    """maxpool3x3s2p1(relu(conv7x7s2p3(half(img)) + bias)) in one launch (csrc/stem.hip) behind a cast pass of our own. img f32
    [N, 3, H, W] (any strides); weight f16/f32 [64, 3, 7, 7]; bias [64] -> f16 [N, 64, Hp, Wp] channels_last."""
    _require_gpu(img, weight, bias)
    n, c, h, w = img.shape
    if img.dtype != torch.float32 or c != 3 or tuple(weight.shape) != (64, 3, 7, 7) or bias.numel() != 64 or h < 8 or w < 8:
        raise ValueError("stem_conv_pool: f32 [N, 3, H, W] image, [64, 3, 7, 7] weight, 64 biases")
    lib = _lib.lib()
    x4 = torch.empty(n, h, w, 4, device=img.device, dtype=torch.float16)
    _lib.check(lib.simpb_image_to_nhwc4_f16(_ptr(x4), _ptr(img), img.stride(0), img.stride(1), img.stride(2), img.stride(3), n, c, h, w,
                                            _stream()), "simpb_image_to_nhwc4_f16")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    out = torch.empty((n, 64, hp, wp), device=img.device, dtype=torch.float16, memory_format=torch.channels_last)
    b16 = bias if bias.dtype == torch.float16 and bias.is_contiguous() else bias.half().contiguous()
    _lib.check(lib.simpb_stem_conv7x7_pool_f16(_ptr(out), _ptr(x4), _ptr(_stem_weight_packed(weight)), _ptr(b16), n, h, w, 64,
                                               _stream()), "simpb_stem_conv7x7_pool_f16")
    return out


def topk_rows(scores, k):
    """(values [bs, k] sorted descending, indices i64 [bs, k]) of scores f32 [bs, n]: torch.topk(sorted=True)
    semantics with ties broken towards the lower index, one launch (csrc/rowops.hip)."""
    _require_gpu(scores)
    if scores.dim() != 2 or scores.shape[1] > 2048:
        raise ValueError("topk_rows takes [bs, n <= 2048] scores")
    scores = scores.contiguous().float()
    bs, n = scores.shape
    values = torch.empty(bs, k, device=scores.device, dtype=torch.float32)
    index = torch.empty(bs, k, device=scores.device, dtype=torch.int32)
    if bs and k:
        _lib.check(_lib.lib().simpb_topk_rows(_ptr(values), _ptr(index), _ptr(scores), bs, n, k, _stream()),
                   "simpb_topk_rows")
    return values, index.long()




def linear_split(x, weight, bias=None):
    """F.linear(x, weight, bias) at fp32-grade accuracy on the FP16 matrix cores (csrc/linear_split.hip):
    both operands split into a leading and a 2^11-scaled trailing half-precision part, three products in
    fp32 accumulators. |x|, |weight| must be below the half-precision range. x [..., K] f32, or f16 (then its trailing part
    is zero and two passes give the same result), weight [N, K]."""
    _require_gpu(x, weight)
    k = x.shape[-1]
    if weight.shape[1] != k or k % 32:
        raise ValueError("linear_split: x [..., K], weight [N, K], K a multiple of 32")
    tag = (weight.data_ptr(), weight._version, str(weight.device))
    hit = getattr(weight, "_simpb_split_lin", None)  # cached on the tensor object: ids / addresses get reused
    if hit is None or hit[0] != tag:
        with torch.no_grad():
            w = weight.detach().float()
            hi = w.half()
            lo = ((w - hi.float()) * 2048.0).half()
        hit = (tag, hi.contiguous(), lo.contiguous())
        weight._simpb_split_lin = hit
    m, n = x.numel() // k, weight.shape[0]
    y = torch.empty(m, n, device=x.device, dtype=torch.float32)
    b = bias.contiguous().float() if bias is not None else None
    if x.dtype == torch.float16 and k % 64 == 0:
        # x is half precision already (no trailing part): two passes (simpb_linear_f16in_split)
        x2 = x.contiguous().reshape(-1, k)
        if m:
            _lib.check(_lib.lib().simpb_linear_f16in_split(_ptr(y), _ptr(x2), _ptr(hit[1]), _ptr(hit[2]), _ptr(b) if b is not None else None,
                                                           m, n, k, _stream()), "simpb_linear_f16in_split")
        return y.reshape(x.shape[:-1] + (n,))
    x2 = x.contiguous().float().reshape(-1, k)
    if m:
        _lib.check(_lib.lib().simpb_linear_f16x3(_ptr(y), _ptr(x2), _ptr(hit[1]), _ptr(hit[2]), _ptr(b) if b is not None else None,
                                                 m, n, k, _stream()), "simpb_linear_f16x3")
    return y.reshape(x.shape[:-1] + (n,))
