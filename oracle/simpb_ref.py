"""ORACLE — test infrastructure only. CPU restatement (plain PyTorch fp32) of the reference's
hybrid 2D/3D decoder hot path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file; the product path (simpb_amd/) must never do so.

Pinned by tests/golden/*.npz, which tools/golden/gen_golden.py captured by running the
reference's own model files in the build container (tests/test_oracle_golden.py). Pieces whose
arithmetic lives in the un-vendored mmcv-full==1.7.1 / mmdet==2.28.2 (requirement.txt:2-3) are
restated from those packages' published semantics and are PARITY UNPINNED: the MultiheadAttention
wrapper, the multi-scale deformable sampler, Scale, bbox_cxcywh_to_xyxy.

Every function cites the reference file:line it follows (paths relative to
/root/reference/projects/mmdet3d_plugin/). Parameters are a flat dict keyed by the reference's
state_dict names (SURVEY.md §8b "Checkpoint compatibility").
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

X, Y, Z, W, L, H, SIN_YAW, COS_YAW, VX, VY, VZ = range(11)  # core/box3d.py:1
CNS, YNS = 0, 1  # core/box3d.py:2


# ----------------------------------------------------------------------------- small pieces
def linear(p, pre, x):
    return F.linear(x, p[pre + ".weight"], p.get(pre + ".bias"))


def layer_norm(p, pre, x):
    return F.layer_norm(x, (x.shape[-1],), p[pre + ".weight"], p[pre + ".bias"], 1e-5)


def linear_relu_ln(p, pre, x, in_loops, out_loops, start=0):
    """models/blocks.py:32-43: out_loops x [in_loops x (Linear, ReLU), LayerNorm] inside an
    nn.Sequential whose first entry has index `start`. Returns (y, next_index)."""
    idx = start
    for _ in range(out_loops):
        for _ in range(in_loops):
            x = F.relu(linear(p, f"{pre}.{idx}", x))
            idx += 2
        x = layer_norm(p, f"{pre}.{idx}", x)
        idx += 1
    return x, idx


def inverse_sigmoid(x, eps=1e-5):
    """models/utils.py:4-8."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def pos2posemb2d(pos, num_pos_feats=128, temperature=10000):
    """models/utils.py:40-63 (2-d branch): order cat(pos_y, pos_x)."""
    pos = pos * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = pos[..., 0, None] / dim_t
    pos_y = pos[..., 1, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=-1).flatten(-2)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=-1).flatten(-2)
    return torch.cat((pos_y, pos_x), dim=-1)


def mha(p, pre, query, key, value, attn_mask=None, num_heads=8):
    """torch.nn.MultiheadAttention forward (batch_first handled by the callers' transposes in
    the reference; here tensors are [bs, N, E]). need_weights=True path: q is scaled before the
    product. `pre` is the prefix of in_proj_weight / in_proj_bias / out_proj.*"""
    e = query.shape[-1]
    w, b = p[pre + ".in_proj_weight"], p[pre + ".in_proj_bias"]
    q = F.linear(query, w[:e], b[:e])
    k = F.linear(key, w[e:2 * e], b[e:2 * e])
    v = F.linear(value, w[2 * e:], b[2 * e:])
    bs, nq, _ = q.shape
    hd = e // num_heads
    q = q.view(bs, nq, num_heads, hd).transpose(1, 2) * math.sqrt(1.0 / hd)
    k = k.view(bs, -1, num_heads, hd).transpose(1, 2)
    v = v.view(bs, -1, num_heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if attn_mask is not None:
        s = s + attn_mask
    a = s.softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(bs, nq, e)
    return linear(p, pre + ".out_proj", o)


def mmcv_mha(p, pre, query, key=None, value=None, query_pos=None, key_pos=None, attn_mask=None):
    """mmcv MultiheadAttention wrapper [mmcv-memory; mirrored by group_attn.py:60-133]:
    identity defaults to the (pre-pos) query, output = identity + attn."""
    if key is None:
        key = query
    if value is None:
        value = key
    identity = query
    if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
        key_pos = query_pos
    if query_pos is not None:
        query = query + query_pos
    if key_pos is not None:
        key = key + key_pos
    return identity + mha(p, pre + ".attn", query, key, value, attn_mask)


def group_mask(n, groups):
    """group_attn.py:104-113: additive float mask, 0 inside a camera group, -inf across."""
    if len(groups) <= 1:
        return None
    m = torch.full((n, n), float("-inf"))
    for s, e in groups:
        m[s:e, s:e] = 0
    return m


# ----------------------------------------------------------------------------- operator: format
def feature_maps_format(feature_maps):
    """ops/__init__.py:63-92: list of [bs, cam, C, H, W] -> [col_feats [bs, sum, C],
    spatial_shape i64[cam, lvl, 2], scale_start_index i64[cam, lvl]]."""
    bs, num_cams = feature_maps[0].shape[:2]
    shapes = [tuple(f.shape[-2:]) for f in feature_maps]
    col = torch.cat([f.reshape(bs, num_cams, f.shape[2], -1) for f in feature_maps], dim=-1)
    col = col.permute(0, 1, 3, 2).flatten(1, 2)
    spatial_shape = torch.tensor([shapes] * num_cams, dtype=torch.int64)
    sizes = (spatial_shape[..., 0] * spatial_shape[..., 1]).flatten()
    start = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)[:-1]]).reshape(num_cams, -1)
    return [col, spatial_shape, start]


# ----------------------------------------------------------------------------- operator: DAF
def deformable_aggregation(feat, spatial_shape, scale_start_index, loc, weights):
    """ops/src/deformable_aggregation_cuda.cu:13-59,129-187 (forward). feat [bs,N,C];
    loc [bs,A,P,cam,2]=(x,y); weights [bs,A,P,cam,lvl,G] -> [bs,A,C]. A sample contributes only
    when 0<x<1 and 0<y<1 (:169-171); pixel = loc*size - 0.5 (:180-181); each of the 4 taps is
    zero outside the map (:35-53); channel c uses weight group c // (C/G) (:149)."""
    bs, _, C = feat.shape
    num_cams, num_lvl = spatial_shape.shape[:2]
    A, P = loc.shape[1:3]
    G = weights.shape[-1]
    out = feat.new_zeros(bs, A, C)
    lx, ly = loc[..., 0], loc[..., 1]
    keep = (lx > 0) & (lx < 1) & (ly > 0) & (ly < 1)
    bidx = torch.arange(bs)[:, None, None].expand(bs, A, P)
    for cam in range(num_cams):
        if not bool(keep[:, :, :, cam].any()):
            continue
        for lvl in range(num_lvl):
            hgt, wid = int(spatial_shape[cam, lvl, 0]), int(spatial_shape[cam, lvl, 1])
            start = int(scale_start_index[cam, lvl])
            fmap = feat[:, start:start + hgt * wid].reshape(bs, hgt, wid, C)
            h_im = ((ly[:, :, :, cam] * hgt).double() - 0.5).float()
            w_im = ((lx[:, :, :, cam] * wid).double() - 0.5).float()
            h_low, w_low = torch.floor(h_im), torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            val = feat.new_zeros(bs, A, P, C)
            for dy, dx, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
                yy, xx = h_low + dy, w_low + dx
                ok = (yy >= 0) & (yy <= hgt - 1) & (xx >= 0) & (xx <= wid - 1) & keep[:, :, :, cam]
                v = fmap[bidx, yy.clamp(0, hgt - 1), xx.clamp(0, wid - 1)]
                val = val + (wt * ok)[..., None] * v
            wg = weights[:, :, :, cam, lvl].repeat_interleave(C // G, dim=-1)
            out = out + (val * wg).sum(2)
    return out


# ----------------------------------------------------------------------------- operator: MSDA
def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """mmcv multi_scale_deformable_attn_pytorch [mmcv-memory]: value [bs,Nv,heads,hd];
    sampling_locations [bs,Nq,heads,lvl,pts,2]; attention_weights [bs,Nq,heads,lvl,pts] ->
    [bs,Nq,heads*hd]; grid_sample(bilinear, zeros, align_corners=False)."""
    bs, _, heads, hd = value.shape
    nq, _, lvls, pts = sampling_locations.shape[1:5]
    vals = value.split([int(h) * int(w) for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(spatial_shapes):
        v = vals[lvl].flatten(2).transpose(1, 2).reshape(bs * heads, hd, int(h), int(w))
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * heads, 1, nq, lvls * pts)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1)
    return out.view(bs, heads * hd, nq).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------- modules
def box3d_encoder(p, pre, box):
    """detection3d/blocks.py:57-74, config: embed_dims [128,32,32,64], mode cat, no output_fc,
    in_loops 1, out_loops 4."""
    pos, _ = linear_relu_ln(p, pre + ".pos_fc", box[..., [X, Y, Z]], 1, 4)
    size, _ = linear_relu_ln(p, pre + ".size_fc", box[..., [W, L, H]], 1, 4)
    yaw, _ = linear_relu_ln(p, pre + ".yaw_fc", box[..., [SIN_YAW, COS_YAW]], 1, 4)
    vel, _ = linear_relu_ln(p, pre + ".vel_fc", box[..., VX:VX + 3], 1, 4)
    return torch.cat([pos, size, yaw, vel], dim=-1)


def box2d_encoder(p, pre, box2d):
    """detection2d/blocks.py:48-63 (with_sin_embed)."""
    return linear_relu_ln(p, pre + ".query_embeddings2d", pos2posemb2d(box2d), 1, 2)[0]


def rotation_mat(anchor):
    """detection3d/blocks.py:202-208 / allocation.py:35-40."""
    r = anchor.new_zeros(anchor.shape[:-1] + (3, 3))
    r[..., 0, 0] = anchor[..., COS_YAW]
    r[..., 0, 1] = -anchor[..., SIN_YAW]
    r[..., 1, 0] = anchor[..., SIN_YAW]
    r[..., 1, 1] = anchor[..., COS_YAW]
    r[..., 2, 2] = 1
    return r


def key_points(p, pre, anchor, feature):
    """detection3d/blocks.py:181-222 (no temporal key points in this config)."""
    bs, n = anchor.shape[:2]
    size = anchor[..., None, [W, L, H]].exp()
    kp = p[pre + ".fix_scale"] * size
    learn = linear(p, pre + ".learnable_fc", feature).reshape(bs, n, -1, 3).sigmoid() - 0.5
    kp = torch.cat([kp, learn * size], dim=-2)
    kp = torch.matmul(rotation_mat(anchor)[:, :, None], kp[..., None]).squeeze(-1)
    return kp + anchor[..., None, [X, Y, Z]]


def project_points(kp, projection_mat, image_wh):
    """blocks.py:198-213 -> [bs, cam, A, P, 2]."""
    ext = torch.cat([kp, torch.ones_like(kp[..., :1])], dim=-1)
    pts = torch.matmul(projection_mat[:, :, None, None], ext[:, None, ..., None]).squeeze(-1)
    pts = pts[..., :2] / torch.clamp(pts[..., 2:3], min=1e-5)
    return pts / image_wh[:, :, None, None]


def dfa_weights(p, pre, feature, anchor_embed, projection_mat, num_cams=6, num_levels=4, num_pts=13, groups=8):
    """blocks.py:164-187 (eval; use_camera_embed)."""
    bs, n = feature.shape[:2]
    feat = feature + anchor_embed
    cam, _ = linear_relu_ln(p, pre + ".camera_encoder", projection_mat[:, :, :3].reshape(bs, num_cams, -1), 1, 2)
    feat = feat[:, :, None] + cam[:, None]
    w = linear(p, pre + ".weights_fc", feat).reshape(bs, n, -1, groups).softmax(dim=-2)
    return w.reshape(bs, n, num_cams, num_levels, num_pts, groups)


def deformable_feature_aggregation(p, pre, feature, anchor, anchor_embed, feature_maps, metas):
    """blocks.py:110-162 with use_deformable_func=True, residual_mode='cat'."""
    bs, n = feature.shape[:2]
    kp = key_points(p, pre + ".kps_generator", anchor, feature)
    w = dfa_weights(p, pre, feature, anchor_embed, metas["projection_mat"])
    num_pts = kp.shape[2]
    pts2d = project_points(kp, metas["projection_mat"], metas["image_wh"]).permute(0, 2, 3, 1, 4)
    pts2d = pts2d.reshape(bs, n, num_pts, -1, 2)
    w = w.permute(0, 1, 4, 2, 3, 5).contiguous()
    col, ss, ssi = feature_maps
    feats = deformable_aggregation(col.float(), ss.int(), ssi.int(), pts2d.contiguous().float(), w.float())
    out = linear(p, pre + ".output_proj", feats)
    return torch.cat([out, feature], dim=-1)


def asymmetric_ffn(p, pre, x):
    """blocks.py:384-393: the identity branch sees the post-LN tensor."""
    x = layer_norm(p, pre + ".pre_norm", x)
    out = linear(p, pre + ".layers.1", F.relu(linear(p, pre + ".layers.0.0", x)))
    return linear(p, pre + ".identity_fc", x) + out


def refine3d(p, pre, feature, anchor, anchor_embed, time_interval, return_cls):
    """detection3d/blocks.py:123-154 (refine_yaw=True, quality estimation on)."""
    feat = feature + anchor_embed
    y, idx = linear_relu_ln(p, pre + ".layers", feat, 2, 2)
    out = linear(p, f"{pre}.layers.{idx}", y) * p[f"{pre}.layers.{idx + 1}.scale"]
    out = out.clone()
    out[..., :8] = out[..., :8] + anchor[..., :8]
    out[..., VX:] = out[..., VX:] / time_interval[:, None, None] + anchor[..., VX:]
    cls = quality = None
    if return_cls:
        c, ci = linear_relu_ln(p, pre + ".cls_layers", feature, 1, 2)
        cls = linear(p, f"{pre}.cls_layers.{ci}", c)
        q, qi = linear_relu_ln(p, pre + ".quality_layers", feat, 1, 2)
        quality = linear(p, f"{pre}.quality_layers.{qi}", q)
    return out, cls, quality


def refine2d(p, pre, feature, anchor2d, anchor2d_embed):
    """detection2d/blocks.py:117-144 (alpha branch on, depth branch off)."""
    y, idx = linear_relu_ln(p, pre + ".layers", feature + anchor2d_embed, 2, 2)
    out = linear(p, f"{pre}.layers.{idx}", y) * p[f"{pre}.layers.{idx + 1}.scale"]
    out = out.clone()
    k = anchor2d.shape[-1]
    out[..., :k] = out[..., :k] + inverse_sigmoid(anchor2d)
    c, ci = linear_relu_ln(p, pre + ".cls_layers", feature, 1, 2)
    cls = linear(p, f"{pre}.cls_layers.{ci}", c)
    a, ai = linear_relu_ln(p, pre + ".alpha_layers", feature, 1, 2)
    alpha = linear(p, f"{pre}.alpha_layers.{ai}", a) * p[f"{pre}.alpha_layers.{ai + 1}.scale"]
    return out.sigmoid(), cls, alpha


def allocation(anchor3d, metas, limit_anchor_size=(35, 35, 10)):
    """allocation.py:27-144 (eval). Returns the reference's 8-tuple minus the unused attn_mask:
    ref_pts2d, ref_depth2d, trans_mask, trans_shape, trans_matrix, center_matrix, query_groups."""
    bs, n = anchor3d.shape[:2]
    proj = metas["projection_mat"]
    num_cams = proj.shape[1]
    img_w, img_h = map(int, metas["image_wh"][0, 0].tolist())
    center = anchor3d[..., :3]
    corners_norm = anchor3d.new_tensor(np.stack(np.unravel_index(np.arange(8), [2] * 3), axis=1)) - 0.5
    size = anchor3d[..., [W, L, H]].exp()
    size = torch.minimum(size, size.new_tensor(limit_anchor_size).view(1, 1, -1))
    corners = size[:, :, None, :] * corners_norm[None, None]
    corners = torch.matmul(rotation_mat(anchor3d)[:, :, None], corners[..., None]).squeeze(-1) + center[:, :, None]
    pts = torch.cat([corners, center[:, :, None]], dim=-2)  # [bs, n, 9, 3]
    pts = torch.cat([pts, torch.ones_like(pts[..., :1])], -1)
    pts2d = torch.matmul(proj[:, None, :, None], pts[:, :, None, :, :, None]).squeeze(-1)  # [bs,n,cam,9,4]
    center2d, corner2d = pts2d[..., -1, :], pts2d[..., :-1, :]
    center_depth, corner_depth = center2d[..., 2:3], corner2d[..., 2:3]
    center2d = center2d[..., :2] / center_depth.clamp(1e-5)
    corner2d = corner2d[..., :2] / corner_depth.clamp(1e-5)
    center_valid = (0 < center2d[..., 0]) & (center2d[..., 0] < img_w) & (0 < center2d[..., 1]) & (center2d[..., 1] < img_h)
    corner_in = (0 < corner2d[..., 0]) & (corner2d[..., 0] < img_w) & (0 < corner2d[..., 1]) & (corner2d[..., 1] < img_h)
    corner_valid = ((corner_depth[..., 0] > 0) & corner_in).any(-1)
    x_min = corner2d[..., 0].min(-1).values.clamp(0, img_w)
    x_max = corner2d[..., 0].max(-1).values.clamp(0, img_w)
    y_min = corner2d[..., 1].min(-1).values.clamp(0, img_h)
    y_max = corner2d[..., 1].max(-1).values.clamp(0, img_h)
    sel = torch.stack([(x_min + x_max) / 2, (y_min + y_max) / 2], dim=-1)
    sel = torch.where(center_valid[..., None], center2d, sel)

    trans_mask = center_valid | corner_valid  # [bs, n, cam]
    trans_shape = trans_mask.sum(1)  # [bs, cam]
    meta = trans_shape.max(0).values
    cum = [0] + meta.cumsum(0).tolist()
    groups = [(cum[i], cum[i + 1]) for i in range(num_cams)]
    n2 = cum[-1]
    ref_pts = anchor3d.new_zeros(bs, n2, 2)
    ref_depth = anchor3d.new_zeros(bs, n2, 1)
    trans = anchor3d.new_zeros(bs, n2, n)
    cmat = anchor3d.new_zeros(bs, n2, n)
    for b in range(bs):
        for cam in range(num_cams):
            idx = torch.nonzero(trans_mask[b, :, cam])[:, 0]  # ascending anchor order (:103-123)
            slots = cum[cam] + torch.arange(len(idx))
            ref_pts[b, slots] = sel[b, idx, cam]
            ref_depth[b, slots] = center_depth[b, idx, cam].abs()
            trans[b, slots, idx] = 1.0
            cmat[b, slots, idx] = center_valid[b, idx, cam].float()
    ref_pts = ref_pts / ref_pts.new_tensor([img_w, img_h])
    return ref_pts, ref_depth, trans_mask, trans_shape, trans, cmat, groups


def qg_msda(p, pre, query, query_pos, anchor2d, groups, enc, num_cams=6, heads=8, lvls=4, pts=4):
    """group_attn.py:146-256 (batch_first, 2-d reference points, residual_mode='cat')."""
    identity = query
    query = query + query_pos
    bs, nq, _ = query.shape
    value = linear(p, pre + ".value_proj", enc["value"])
    nv = value.shape[1]
    value = value.view(bs, num_cams, nv, heads, -1)
    off = linear(p, pre + ".sampling_offsets", query).view(bs, nq, heads, lvls, pts, 2)
    aw = linear(p, pre + ".attention_weights", query).view(bs, nq, heads, lvls * pts).softmax(-1)
    aw = aw.view(bs, nq, heads, lvls, pts)
    ss = enc["spatial_shapes"]
    norm = torch.stack([ss[..., 1], ss[..., 0]], -1)
    ref = anchor2d[..., :2].unsqueeze(2)  # simpb_head.py:523
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    outs = []
    for i, (s, e) in enumerate(groups):
        if e - s > 0:
            outs.append(ms_deform_attn(value[:, i].contiguous(), ss, loc[:, s:e].contiguous(), aw[:, s:e].contiguous()))
    out = linear(p, pre + ".output_proj", torch.cat(outs, dim=1))
    return torch.cat([out, identity], dim=-1)


def aggregation(p, pre, q2d, pos2d, q3d, pos3d, trans, cmat, graph_model):
    """aggregation.py:10-101 (reweight, with_pos, self_attn; eval so no dn branch)."""
    param = torch.cat([q2d, cmat.sum(-1, keepdim=True)], dim=-1)
    alpha = torch.sigmoid(linear(p, pre + ".reweight.alpha.0", F.relu(linear(p, pre + ".reweight.reduce.0", param))))
    rw = (trans * alpha).permute(0, 2, 1)
    div = torch.clamp(rw.sum(-1, keepdim=True), 1e-5)
    q3d = q3d + torch.matmul(rw, q2d) / div
    pos3d = pos3d + torch.matmul(rw, pos2d) / div
    return graph_model(pre + ".self_attn", q3d, query_pos=pos3d), pos3d


# ----------------------------------------------------------------------------- instance bank
def bank_topk(conf, k, *inputs):
    """instance_bank.py:13-20."""
    conf, idx = torch.topk(conf, k, dim=1)
    outs = [torch.gather(x, 1, idx[..., None].expand(-1, -1, x.shape[-1])) for x in inputs]
    return conf, outs


def anchor_projection(anchor, t_src2dst, time_interval):
    """detection3d/blocks.py:248-280, bug-for-bug: the yaw pair goes in as [cos, sin] and is
    written back without re-ordering (:271-278), so slot 6 (SIN) receives the rotated cos."""
    t = t_src2dst[:, None].to(anchor.dtype)
    vel = anchor[..., VX:]
    center = anchor[..., [X, Y, Z]] - vel * time_interval[:, None, None]
    center = torch.matmul(t[..., :3, :3], center[..., None]).squeeze(-1) + t[..., :3, 3]
    yaw = torch.matmul(t[..., :2, :2], anchor[..., [COS_YAW, SIN_YAW], None]).squeeze(-1)
    vel = torch.matmul(t[..., :3, :3], vel[..., None]).squeeze(-1)
    return torch.cat([center, anchor[..., [W, L, H]], yaw, vel], dim=-1)


class InstanceBank:
    """instance_bank.py:23-196, eval path."""

    def __init__(self, p, pre, num_anchor, num_temp, default_dt=0.5, decay=0.6, max_dt=2):
        self.anchor = p[pre + ".anchor"]
        self.feature = p[pre + ".instance_feature"]
        self.num_anchor, self.num_temp = num_anchor, num_temp
        self.default_dt, self.decay, self.max_dt = default_dt, decay, max_dt
        self.reset()

    def reset(self):
        self.cached_feature = self.cached_anchor = self.metas = self.mask = None
        self.confidence = self.temp_confidence = self.instance_id = None
        self.prev_id = 0

    def get(self, bs, metas):
        feat = self.feature[None].repeat(bs, 1, 1)
        anchor = self.anchor[None].repeat(bs, 1, 1)
        if self.cached_anchor is not None and bs == self.cached_anchor.shape[0]:
            dt = (metas["timestamp"] - self.metas["timestamp"]).to(feat.dtype)
            self.mask = torch.abs(dt) <= self.max_dt
            t = np.stack([m["T_global_inv"] @ self.metas["img_metas"][i]["T_global"]
                          for i, m in enumerate(metas["img_metas"])])
            self.cached_anchor = anchor_projection(self.cached_anchor, self.cached_anchor.new_tensor(t), -dt)
            dt = torch.where((dt != 0) & self.mask, dt, dt.new_tensor(self.default_dt))
        else:
            self.reset()
            dt = feat.new_tensor([self.default_dt] * bs)
        return feat, anchor, self.cached_feature, self.cached_anchor, dt

    def update(self, feat, anchor, cls):
        if self.cached_feature is None:
            return feat, anchor
        n = self.num_anchor - self.num_temp
        _, (sf, sa) = bank_topk(cls.max(dim=-1).values, n, feat, anchor)
        sf = torch.cat([self.cached_feature, sf], dim=1)
        sa = torch.cat([self.cached_anchor, sa], dim=1)
        feat = torch.where(self.mask[:, None, None], sf, feat)
        anchor = torch.where(self.mask[:, None, None], sa, anchor)
        if self.instance_id is not None:
            self.instance_id = torch.where(self.mask[:, None], self.instance_id, self.instance_id.new_tensor(-1))
        return feat, anchor

    def cache(self, feat, anchor, cls, metas):
        self.metas = metas
        conf = cls.max(dim=-1).values.sigmoid()
        if self.confidence is not None:
            conf[:, :self.num_temp] = torch.maximum(self.confidence * self.decay, conf[:, :self.num_temp])
        self.temp_confidence = conf
        self.confidence, (self.cached_feature, self.cached_anchor) = bank_topk(conf, self.num_temp, feat, anchor)

    def get_instance_id(self, cls, threshold=None):
        conf = cls.max(dim=-1).values.sigmoid()
        ids = conf.new_full(conf.shape, -1).long()
        if self.instance_id is not None and self.instance_id.shape[0] == ids.shape[0]:
            ids[:, :self.instance_id.shape[1]] = self.instance_id
        mask = ids < 0
        if threshold is not None:
            mask = mask & (conf >= threshold)
        n_new = int(mask.sum())
        ids[torch.where(mask)] = torch.arange(n_new).to(ids) + self.prev_id
        self.prev_id += n_new
        kept = bank_topk(self.temp_confidence, self.num_temp, ids[..., None])[1][0].squeeze(-1)
        self.instance_id = F.pad(kept, (0, self.num_anchor - self.num_temp), value=-1)
        return ids


# ----------------------------------------------------------------------------- decoder
def decode_box(box):
    """detection3d/decoder.py:23-34."""
    yaw = torch.atan2(box[:, SIN_YAW], box[:, COS_YAW])
    return torch.cat([box[:, [X, Y, Z]], box[:, [W, L, H]].exp(), yaw[:, None], box[:, VX:]], dim=-1)


def decode_box2d(box, aug_config):
    """detection3d/decoder.py:36-51; bbox_cxcywh_to_xyxy is mmdet's [mmcv-memory]."""
    crop, scale = aug_config["crop"], aug_config["resize"]
    cw, ch = crop[2] - crop[0], crop[3] - crop[1]
    cx, cy, w, h = box.unbind(-1)
    box = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
    box[..., 0::2] = (box[..., 0::2] * cw).clamp(0, cw)
    box[..., 1::2] = (box[..., 1::2] * ch).clamp(0, ch) + crop[1]
    return box / scale


def decode_with2d(cls, box, instance_id, quality, cls2d, box2d, trans, groups, aug_configs, num_output=300,
                  score_threshold=None):
    """detection3d/decoder.py:124-252 with squeeze_cls (instance ids present), association on."""
    scores, cls_ids = cls.sigmoid().max(dim=-1)
    bs = scores.shape[0]
    scores, indices = scores.topk(num_output, dim=1, sorted=True)
    origin = scores.clone()
    cns = torch.gather(quality[..., CNS], 1, indices)
    scores = scores * cns.sigmoid()
    scores, order = torch.sort(scores, dim=1, descending=True)
    indices = torch.gather(indices, 1, order)
    trans_t = trans.permute(0, 2, 1)
    out = []
    for i in range(bs):
        t = trans_t[i, indices[i]]
        idx2d = torch.where(t.any(0))[0]
        t = torch.index_select(t, 1, idx2d)
        camidx, new_groups = [], []
        for cam, (s, e) in enumerate(groups):
            part = torch.where((s <= idx2d) & (idx2d < e))[0]
            if len(part) > 0:
                g = (int(part[0]), int(part[-1]) + 1)
            elif new_groups:
                g = (new_groups[-1][-1], new_groups[-1][-1])
            else:
                g = (0, 0)
            camidx.append(torch.ones(len(part)) * cam)
            new_groups.append(g)
        groups = new_groups  # the reference re-binds the loop variable (:216)
        s2d, l2d = cls2d[i, idx2d].sigmoid().max(dim=-1)
        out.append(dict(
            boxes_3d=decode_box(box[i, indices[i]]), scores_3d=scores[i], labels_3d=cls_ids[i][indices[i]],
            boxes_2d=decode_box2d(box2d[i, idx2d], aug_configs[0]), scores_2d=s2d, labels_2d=l2d,
            camidx_2d=torch.cat(camidx), trans_matrix=t, query_groups=new_groups, cls_scores=origin[i],
            instance_ids=instance_id[i, indices[i]],
        ))
    return out


# ----------------------------------------------------------------------------- the head
class OracleHead:
    """simpb_head.py:31-747 (eval path) + post_process :1089-1123."""

    def __init__(self, params, operation_order, num_anchor=900, num_temp=600, num_output=300,
                 num_single_frame_decoder=1, trace=None):
        self.p = {k: v.float() if v.is_floating_point() else v for k, v in params.items()}
        self.ops = list(operation_order)
        self.bank = InstanceBank(self.p, "instance_bank", num_anchor, num_temp)
        self.num_output = num_output
        self.nsfd = num_single_frame_decoder
        self.trace = trace

    def _t(self, name, val):
        if self.trace is None:
            return
        if isinstance(val, (list, tuple)):
            for k, v in enumerate([v for v in val if v is not None]):
                self.trace.add(f"{name}.{k}", v)
        else:
            self.trace.add(name, val)

    def anchor_encoder(self, anchor):
        out = box3d_encoder(self.p, "anchor_encoder", anchor)
        self._t("anchor_encoder", out)
        return out

    def graph_model(self, pre, query, key=None, value=None, query_pos=None, key_pos=None, lname=None):
        """simpb_head.py:298-310 (decouple_attn)."""
        q = torch.cat([query, query_pos], dim=-1)
        k = torch.cat([key, key_pos], dim=-1) if key is not None else None
        v = linear(self.p, "fc_before", value) if value is not None else None
        out = mmcv_mha(self.p, pre, q, k, v)
        if lname:
            self._t(lname + ".0", out)
        out = linear(self.p, "fc_after", out)
        self._t("fc_after", out)
        return out

    def forward(self, feature_maps, metas):
        p = self.p
        bs = feature_maps[0].shape[0]
        feat, anchor, temp_feat, temp_anchor, dt = self.bank.get(bs, metas)
        embed = self.anchor_encoder(anchor)
        temp_embed = self.anchor_encoder(temp_anchor) if temp_anchor is not None else None
        col, ss, ssi = feature_maps
        num_cams = ss.shape[0]
        enc = dict(value=col.reshape(bs, num_cams, -1, col.shape[-1]).flatten(0, 1),
                   spatial_shapes=ss[0].long())  # simpb_head.py:281-292
        pred, clss, qual, pred2d, cls2d, trans_list, groups_list = [], [], [], [], [], [], []
        temp_attn = feat
        for i, op in enumerate(self.ops):
            pre, ln = f"layers.{i}", f"L{i:02d}.{op}"
            if op == "norm":
                feat = layer_norm(p, pre, feat)
                self._t(ln + ".0", feat)
            elif op == "ffn":
                feat = asymmetric_ffn(p, pre, feat)
                self._t(ln + ".0", feat)
            elif op == "allocation":
                anchor2d, depth2d, tmask, tshape, trans, cmat, groups = allocation(anchor, metas)
                if self.trace is not None:
                    self.trace.add(ln + ".ref_pts2d", anchor2d)
                    self.trace.add(ln + ".ref_depth2d", depth2d)
                    self.trace.add(ln + ".trans_mask", tmask)
                    self.trace.add(ln + ".trans_shape", tshape)
                    self.trace.add(ln + ".q2a", torch.where(trans.sum(-1) > 0, trans.argmax(-1), -1).to(torch.int32))
                    self.trace.add(ln + ".is_center", cmat.sum(-1).to(torch.int32))
                    self.trace.add(ln + ".query_groups", torch.tensor(groups, dtype=torch.int32))
                feat = torch.matmul(trans, feat)  # simpb_head.py:438
                embed2d = box2d_encoder(p, "anchor_encoder2d", anchor2d)
                self._t("anchor_encoder2d", embed2d)
            elif op == "qg_self_attn":
                # graph_model2d (simpb_head.py:312-321) around group_attn.py:60-133
                q = torch.cat([feat, embed2d], dim=-1)
                out = q + torch.nan_to_num(mha(p, pre + ".attn", q, q, linear(p, "fc_before2d", feat),
                                               group_mask(q.shape[1], groups)))
                self._t(ln + ".0", out)
                feat = linear(p, "fc_after2d", out)
                self._t("fc_after2d", feat)
            elif op == "qg_cross_attn":
                feat = qg_msda(p, pre, feat, embed2d, anchor2d, groups, enc, num_cams)
                self._t(ln + ".0", feat)
            elif op == "refine2d":
                anchor2d, c2d, alpha = refine2d(p, pre, feat, anchor2d, embed2d)
                self._t(ln, [anchor2d, c2d, alpha])
                pred2d.append(anchor2d)
                cls2d.append(c2d)
                trans_list.append(trans)
                groups_list.append(groups)
            elif op == "aggregation":
                feat, embed = aggregation(p, pre, feat, embed2d, temp_attn, embed, trans, cmat, self.graph_model)
                self._t(ln, [feat, embed, anchor])
            elif op == "gnn":
                feat = self.graph_model(pre, feat, value=feat, query_pos=embed, lname=ln)
            elif op == "temp_gnn":
                feat = self.graph_model(pre, feat, temp_feat, temp_feat, query_pos=embed, key_pos=temp_embed, lname=ln)
                temp_attn = feat
            elif op == "deformable":
                feat = deformable_feature_aggregation(p, pre, feat, anchor, embed, feature_maps, metas)
                self._t(ln + ".0", feat)
            elif op == "refine3d":
                ret = len(pred) == self.nsfd - 1 or i == len(self.ops) - 1
                anchor, c, q = refine3d(p, pre, feat, anchor, embed, dt, ret)
                self._t(ln, [anchor, c, q])
                pred.append(anchor)
                clss.append(c)
                qual.append(q)
                if len(pred) == self.nsfd:
                    feat, anchor = self.bank.update(feat, anchor, c)
                if i != len(self.ops) - 1:
                    embed = self.anchor_encoder(anchor)
                if len(pred) > self.nsfd and temp_embed is not None:
                    temp_embed = embed[:, :self.bank.num_temp]
            else:
                raise NotImplementedError(op)
        self.bank.cache(feat, anchor, c, metas)
        ids = self.bank.get_instance_id(c)
        return dict(prediction=pred, classification=clss, quality=qual, prediction2d=pred2d, classification2d=cls2d,
                    ref_trans_matrix_list=trans_list, ref_query_groups_list=groups_list, instance_id=ids)

    def post_process(self, outs, metas):
        aug = [m["aug_config"] for m in metas["img_metas"]]
        return decode_with2d(outs["classification"][-1], outs["prediction"][-1], outs["instance_id"],
                             outs["quality"][-1], outs["classification2d"][-1], outs["prediction2d"][-1],
                             outs["ref_trans_matrix_list"][-1], outs["ref_query_groups_list"][-1], aug,
                             self.num_output)
