"""models/allocation.py of the reference: DynamicQueryAllocation, computed by the HIP kernels of
csrc/alloc.hip. The one host round trip the reference also has (:94, `.tolist()` of the per-camera
counts) is kept: it sizes the 2D query set."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .ops import _ptr, _require_gpu, _stream
from .registry import PLUGIN_LAYERS


# routes.alloc_static_fused: with a fixed capacity the allocation steps run as three launches (simpb_alloc_static); off:
# five, one per step (the route the exact-size mode always takes), also the cross-check in tests.
from . import routes


class Allocation2D:
    """Index form of one allocation: what the reference spreads over ref_trans_matrix /
    ref_center_matrix / query_groups."""

    __slots__ = ("q2a", "is_center", "a2q", "query_cam", "query_groups", "num_anchor", "count", "group_start",
                 "overflow", "streams")

    def __init__(self):
        # 0: the reference's batch layout [bs, N2] (groups padded to the max over the batch). bs > 0: a batch of that many
        # INDEPENDENT streams as one flat slot array [1, bs * capacity] over bs * cams groups (csrc/alloc.hip:
        # alloc_scatter_ragged_kernel); q2a then holds flat anchor indices b * num_anchor + a
        self.streams = 0

    def dense(self):
        """The reference's (trans_matrix, center_matrix) one-hot f32 [bs, N2, N3] (allocation.py:128-142)."""
        if self.streams:
            raise NotImplementedError("a batch of independent streams has no [bs, N2, N3] matrix form")
        bs, n2 = self.q2a.shape
        trans = torch.zeros(bs, n2, self.num_anchor + 1, device=self.q2a.device)
        idx = torch.where(self.q2a >= 0, self.q2a, self.num_anchor).long()
        trans.scatter_(2, idx[..., None], 1.0)
        trans = trans[..., : self.num_anchor].contiguous()
        return trans, trans * self.is_center[..., None].float()


@PLUGIN_LAYERS.register_module()
class DynamicQueryAllocation(nn.Module):
    def __init__(self, with_attn_mask=False, with_project_wh=False, limit_anchor_size=[35, 35, 10],
                 limit_corners_num=[100] * 6):
        super().__init__()
        if with_attn_mask:
            raise NotImplementedError("with_attn_mask is off in the SimPB configs (with_allocate_attn_mask=False)")
        self.with_attn_mask = with_attn_mask
        self.with_project_wh = with_project_wh
        self.limit_anchor_size = limit_anchor_size
        self.limit_corners_num = limit_corners_num
        self.last = None

    def forward(self, anchor3d, metas, dense=True, capacity=None, overflow_out=None, independent=False):
        """Returns the reference's 8-tuple (allocation.py:144). With dense=True the two one-hot
        matrices are materialised from the index form; with dense=False their places hold None
        and callers use `self.last` (an Allocation2D) instead."""
        if independent:
            alloc, ref_pts2d, ref_depth2d = self.allocate_independent(anchor3d, metas, capacity, overflow_out)
            if dense:
                raise ValueError("independent streams: index form only (dense=False)")
            return ref_pts2d, ref_depth2d, None, None, None, None, None, None
        alloc, ref_pts2d, ref_depth2d, trans_mask, trans_shape = self.allocate(anchor3d, metas, capacity, overflow_out)
        trans, center = alloc.dense() if dense else (None, None)
        return ref_pts2d, ref_depth2d, trans_mask, trans_shape, trans, center, alloc.query_groups, None

    def allocate_independent(self, anchor3d, metas, capacity, overflow_out=None):
        """The batch as `bs` independent camera streams: every stream keeps the 2D set a batch of one gives
        it, at most `capacity` slots each; one flat slot array [1, bs * capacity] over bs * cams groups, live slots first
        (Allocation2D.streams = bs). Static shapes only."""
        if self.training:
            raise NotImplementedError("training-time corner sampling (allocation.py:85-87) is not on this path")
        if capacity is None:
            raise ValueError("independent streams need a static capacity (slots per stream)")
        _require_gpu(anchor3d)
        lib = _lib.lib()
        anchor3d = anchor3d.contiguous().float()
        proj = metas["projection_mat"].contiguous().float()
        bs, num_anchor = anchor3d.shape[:2]
        cams = proj.shape[1]
        if anchor3d.shape[-1] != 11 or tuple(proj.shape) != (bs, cams, 4, 4):
            raise ValueError("anchor3d must be [bs, N, 11] and projection_mat [bs, cams, 4, 4]")
        wh = metas.get("image_wh_host")
        if wh is None:
            wh = tuple(int(v) for v in metas["image_wh"][0, 0].tolist())
        img_w, img_h = float(wh[0]), float(wh[1])
        dev = anchor3d.device
        lw, ll, lh = (float(v) for v in self.limit_anchor_size)
        slots = bs * int(capacity)
        flag = torch.empty(bs, cams, num_anchor, dtype=torch.uint8, device=dev)
        sel_xy = torch.empty(bs, cams, num_anchor, 2, device=dev)
        depth = torch.empty(bs, cams, num_anchor, device=dev)
        overflow = overflow_out if overflow_out is not None else torch.empty(1, dtype=torch.int32, device=dev)
        if overflow.dtype != torch.int32 or overflow.numel() != 1 or overflow.device != dev:
            raise ValueError("overflow_out must be one i32 element on the anchors' device")
        out = Allocation2D()
        out.streams = bs
        out.count = torch.empty(bs, cams, dtype=torch.int32, device=dev)
        order = torch.empty(bs, cams, num_anchor, dtype=torch.int32, device=dev)
        out.group_start = torch.empty(bs * cams + 1, dtype=torch.int32, device=dev)
        ref_pts2d = torch.empty(1, slots, 2, device=dev)
        ref_depth2d = torch.empty(1, slots, 1, device=dev)
        out.q2a = torch.empty(1, slots, dtype=torch.int32, device=dev)
        out.is_center = torch.empty(1, slots, dtype=torch.int32, device=dev)
        out.a2q = torch.empty(bs, num_anchor, cams, dtype=torch.int32, device=dev)
        out.query_cam = torch.empty(slots, dtype=torch.int32, device=dev)
        out.query_groups = None
        out.num_anchor = num_anchor
        out.overflow = overflow
        _lib.check(lib.simpb_alloc_ragged(_ptr(flag), _ptr(sel_xy), _ptr(depth), _ptr(out.count), _ptr(order),
                                          _ptr(out.group_start), _ptr(overflow), _ptr(ref_pts2d), _ptr(ref_depth2d),
                                          _ptr(out.q2a), _ptr(out.is_center), _ptr(out.a2q), _ptr(out.query_cam), _ptr(anchor3d),
                                          _ptr(proj), bs, num_anchor, cams, int(capacity), img_w, img_h, lw, ll, lh, _stream()),
                   "simpb_alloc_ragged")
        self.last = out
        return out, ref_pts2d, ref_depth2d

    def allocate(self, anchor3d, metas, capacity=None, overflow_out=None):
        """capacity=None: size the 2D set exactly (one count readback, like allocation.py:94).
        capacity=N: static shapes, no host round trip; the group table stays on the device, slots
        past the last group carry query_cam = -1, and `overflow` (i32 [1]; `overflow_out` when the caller keeps the
        flags of a frame's layers in one tensor) flags a set that did not fit."""
        if self.training:
            raise NotImplementedError("training-time corner sampling (allocation.py:85-87) is not on this path")
        _require_gpu(anchor3d)
        lib = _lib.lib()
        anchor3d = anchor3d.contiguous().float()
        proj = metas["projection_mat"].contiguous().float()
        bs, num_anchor = anchor3d.shape[:2]
        cams = proj.shape[1]
        if anchor3d.shape[-1] != 11 or tuple(proj.shape) != (bs, cams, 4, 4):
            raise ValueError("anchor3d must be [bs, N, 11] and projection_mat [bs, cams, 4, 4]")
        wh = metas.get("image_wh_host")
        if wh is None:  # allocation.py:32 reads it from the device tensor every call
            wh = tuple(int(v) for v in metas["image_wh"][0, 0].tolist())
        img_w, img_h = float(wh[0]), float(wh[1])
        dev = anchor3d.device
        flag = torch.empty(bs, cams, num_anchor, dtype=torch.uint8, device=dev)
        sel_xy = torch.empty(bs, cams, num_anchor, 2, device=dev)
        depth = torch.empty(bs, cams, num_anchor, device=dev)
        lw, ll, lh = (float(v) for v in self.limit_anchor_size)
        st = _stream()
        count = torch.empty(bs, cams, dtype=torch.int32, device=dev)
        order = torch.empty(bs, cams, num_anchor, dtype=torch.int32, device=dev)
        static = capacity is not None and routes.R.alloc_static_fused and cams <= 8
        if not static:
            _lib.check(lib.simpb_alloc_project(_ptr(flag), _ptr(sel_xy), _ptr(depth), _ptr(anchor3d), _ptr(proj), bs,
                                               num_anchor, cams, img_w, img_h, lw, ll, lh, st), "simpb_alloc_project")
            _lib.check(lib.simpb_alloc_compact(_ptr(count), _ptr(order), _ptr(flag), bs, num_anchor, cams, st),
                       "simpb_alloc_compact")
        overflow = None
        if capacity is None:
            meta = count.max(dim=0).values.tolist()  # the one device->host sync (allocation.py:91-94)
            cum = [0]
            for c in meta:
                cum.append(cum[-1] + int(c))
            n2 = cum[-1]
            group_start = torch.tensor(cum, dtype=torch.int32).to(dev, non_blocking=True)
        else:
            cum, n2 = None, int(capacity)
            group_start = torch.empty(cams + 1, dtype=torch.int32, device=dev)
            overflow = overflow_out if overflow_out is not None else torch.empty(1, dtype=torch.int32, device=dev)
            if overflow.dtype != torch.int32 or overflow.numel() != 1 or overflow.device != dev:
                raise ValueError("overflow_out must be one i32 element on the anchors' device")
            if not static:
                _lib.check(lib.simpb_alloc_group_start(_ptr(group_start), _ptr(overflow), _ptr(count), bs, cams, n2, st),
                           "simpb_alloc_group_start")
        ref_pts2d = torch.empty(bs, n2, 2, device=dev)
        ref_depth2d = torch.empty(bs, n2, 1, device=dev)
        out = Allocation2D()
        out.q2a = torch.empty(bs, n2, dtype=torch.int32, device=dev)
        out.is_center = torch.empty(bs, n2, dtype=torch.int32, device=dev)
        out.a2q = torch.empty(bs, num_anchor, cams, dtype=torch.int32, device=dev)
        out.query_cam = torch.empty(n2, dtype=torch.int32, device=dev)
        out.query_groups = [(cum[i], cum[i + 1]) for i in range(cams)] if cum is not None else None
        out.num_anchor = num_anchor
        out.count = count
        out.group_start = group_start
        out.overflow = overflow
        if static:
            # fixed capacity: nothing returns to the host between the steps: fill and group table ride in steps 1 and 3
            _lib.check(lib.simpb_alloc_static(_ptr(flag), _ptr(sel_xy), _ptr(depth), _ptr(count), _ptr(order), _ptr(group_start),
                                              _ptr(overflow), _ptr(ref_pts2d), _ptr(ref_depth2d), _ptr(out.q2a),
                                              _ptr(out.is_center), _ptr(out.a2q), _ptr(out.query_cam), _ptr(anchor3d), _ptr(proj),
                                              bs, num_anchor, cams, n2, img_w, img_h, lw, ll, lh, st), "simpb_alloc_static")
        else:
            _lib.check(lib.simpb_alloc_scatter(_ptr(ref_pts2d), _ptr(ref_depth2d), _ptr(out.q2a), _ptr(out.is_center),
                                               _ptr(out.a2q), _ptr(out.query_cam), _ptr(group_start), _ptr(count),
                                               _ptr(order), _ptr(flag), _ptr(sel_xy), _ptr(depth), bs, num_anchor, cams, n2,
                                               img_w, img_h, st), "simpb_alloc_scatter")
        self.last = out
        if capacity is not None:
            # static mode: the mask / count views of the reference's tuple (allocation.py:144) are only consumed by
            # the variable-shape decode; the static decode works from `out`, so they are not materialised
            return out, ref_pts2d, ref_depth2d, None, None
        trans_mask = (flag != 0).permute(0, 2, 1)
        return out, ref_pts2d, ref_depth2d, trans_mask, count.long()


def gather_rows(src, q2a):
    """out[b, s] = src[b, q2a[b, s]] (zeros for pads): torch.matmul(ref_trans_matrix, feature) at
    simpb_head.py:438 without the one-hot matrix."""
    _require_gpu(src, q2a)
    src = src.contiguous().float()
    bs, num_anchor, c = src.shape
    n2 = q2a.shape[1]
    if q2a.dtype != torch.int32 or not q2a.is_contiguous() or q2a.shape[0] != bs or c % 4:
        raise ValueError("q2a must be contiguous i32 [bs, N2]; channels a multiple of 4")
    out = torch.empty(bs, n2, c, device=src.device)
    if n2:
        _lib.check(_lib.lib().simpb_gather_rows(_ptr(out), _ptr(src), _ptr(q2a), bs, num_anchor, n2, c, _stream()),
                   "simpb_gather_rows")
    return out


def aggregate_2d_to_3d(q3d, pos3d, q2d, pos2d, alpha, a2q, hidden=None, alpha_fc=None):
    """aggregation.py:30-35,88-89 as one kernel over the (anchor, cam) -> slot table. alpha: per-slot weights [bs, N2, 1],
    or None with `hidden` [bs, N2, k] and `alpha_fc` (the Linear(k, 1) of ReWeight.alpha): the kernel then computes
    sigmoid(alpha_fc(hidden)) itself (aggregation.py:23-24)."""
    _require_gpu(q3d, pos3d, q2d, pos2d, a2q)
    q3d, pos3d, q2d, pos2d = (t.contiguous().float() for t in (q3d, pos3d, q2d, pos2d))
    if alpha is None:
        return _aggregate_alpha(q3d, pos3d, q2d, pos2d, a2q, hidden, alpha_fc)
    alpha = alpha.contiguous().float()
    bs, num_anchor, c = q3d.shape
    n2 = q2d.shape[1]
    cams = a2q.shape[-1]
    if (tuple(pos3d.shape) != (bs, num_anchor, c) or tuple(q2d.shape) != (bs, n2, c) or tuple(pos2d.shape) != (bs, n2, c)
            or alpha.numel() != bs * n2 or tuple(a2q.shape) != (bs, num_anchor, cams) or a2q.dtype != torch.int32
            or not a2q.is_contiguous() or c % 4):
        raise ValueError("aggregate_2d_to_3d: inconsistent shapes")
    if n2 == 0:
        return q3d.clone(), pos3d.clone()
    out_q, out_pos = torch.empty_like(q3d), torch.empty_like(pos3d)
    _lib.check(_lib.lib().simpb_aggregate_2d_to_3d(_ptr(out_q), _ptr(out_pos), _ptr(q3d), _ptr(pos3d), _ptr(q2d),
                                                   _ptr(pos2d), _ptr(alpha), _ptr(a2q), bs, num_anchor, cams, n2, c,
                                                   _stream()), "simpb_aggregate_2d_to_3d")
    return out_q, out_pos


def _aggregate_alpha(q3d, pos3d, q2d, pos2d, a2q, hidden, alpha_fc):
    bs, num_anchor, c = q3d.shape
    n2, cams = q2d.shape[1], a2q.shape[-1]
    hidden = hidden.contiguous().float()
    k = hidden.shape[-1]
    w = alpha_fc.weight.reshape(-1).float().contiguous()
    if (tuple(pos3d.shape) != (bs, num_anchor, c) or tuple(q2d.shape) != (bs, n2, c) or tuple(pos2d.shape) != (bs, n2, c)
            or tuple(hidden.shape) != (bs, n2, k) or w.numel() != k or tuple(a2q.shape) != (bs, num_anchor, cams)
            or a2q.dtype != torch.int32 or not a2q.is_contiguous() or c % 4 or k % 4 or n2 == 0):
        raise ValueError("aggregate_2d_to_3d: inconsistent shapes")
    out_q, out_pos = torch.empty_like(q3d), torch.empty_like(pos3d)
    _lib.check(_lib.lib().simpb_aggregate_2d_to_3d_alpha(
        _ptr(out_q), _ptr(out_pos), _ptr(q3d), _ptr(pos3d), _ptr(q2d), _ptr(pos2d), None, _ptr(a2q), _ptr(hidden), k, k, _ptr(w),
        _ptr(alpha_fc.bias) if alpha_fc.bias is not None else None, bs, num_anchor, cams, n2, c, _stream()),
        "simpb_aggregate_2d_to_3d_alpha")
    return out_q, out_pos
