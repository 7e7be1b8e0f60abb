"""models/detection3d/{blocks,decoder}.py of the reference: 3D box encoder, refinement head,
key-point generator and the box decoder. The training-only target/loss classes of that package
are registered as inert stand-ins in training_stubs.py."""
import torch
import torch.nn as nn

from .box3d import *  # noqa: F401,F403
from .box3d import COS_YAW, H, L, SIN_YAW, VX, W, X, Y, Z, CNS

# X,Y,Z / W,L,H / SIN,COS are contiguous index runs (core/box3d.py:1): plain slices are used instead
# of list indexing, which would build an index tensor on the host for every call.
assert (X, Y, Z, W, L, H, SIN_YAW, COS_YAW) == tuple(range(8))
from .layers import BaseModule, Linear, Scale, bias_init_with_prob, linear_relu_ln
from .registry import BBOX_CODERS, PLUGIN_LAYERS, POSITIONAL_ENCODING

# routes.fused_decode: decode_static_device builds its two records with csrc/decode.hip (two launches); off: the
# PyTorch statement of the same arithmetic (~40 launches), kept as the cross-check in tests/test_gpu_runner.py.
from . import routes

__all__ = ["SparseBox3DRefinementModule", "SparseBox3DKeyPointsGenerator", "SparseBox3DEncoder", "SparseBox3DDecoder"]


@POSITIONAL_ENCODING.register_module()
class SparseBox3DEncoder(BaseModule):
    """detection3d/blocks.py:23-74."""

    def __init__(self, embed_dims, vel_dims=3, mode="add", output_fc=True, in_loops=1, out_loops=2):
        super().__init__()
        assert mode in ["add", "cat"]
        self.embed_dims = embed_dims
        self.vel_dims = vel_dims
        self.mode = mode

        def embedding_layer(input_dims, output_dims):
            return nn.Sequential(*linear_relu_ln(output_dims, in_loops, out_loops, input_dims))

        if not isinstance(embed_dims, (list, tuple)):
            embed_dims = [embed_dims] * 5
        self.pos_fc = embedding_layer(3, embed_dims[0])
        self.size_fc = embedding_layer(3, embed_dims[1])
        self.yaw_fc = embedding_layer(2, embed_dims[2])
        if vel_dims > 0:
            self.vel_fc = embedding_layer(self.vel_dims, embed_dims[3])
        self.output_fc = embedding_layer(embed_dims[-1], embed_dims[-1]) if output_fc else None

    def forward(self, box_3d):
        if box_3d.is_cuda and self.mode == "cat" and self.output_fc is None and self.vel_dims > 0:
            return self._forward_fused(box_3d)
        pos_feat = self.pos_fc(box_3d[..., X:Z + 1])
        size_feat = self.size_fc(box_3d[..., W:H + 1])
        yaw_feat = self.yaw_fc(box_3d[..., SIN_YAW:COS_YAW + 1])
        if self.mode == "add":
            output = pos_feat + size_feat + yaw_feat
        else:
            output = torch.cat([pos_feat, size_feat, yaw_feat], dim=-1)
        if self.vel_dims > 0:
            vel_feat = self.vel_fc(box_3d[..., VX: VX + self.vel_dims])
            output = output + vel_feat if self.mode == "add" else torch.cat([output, vel_feat], dim=-1)
        if self.output_fc is not None:
            output = self.output_fc(output)
        return output


    def _forward_fused(self, box_3d):
        """The 4 branches (48 Linear/ReLU/LN modules) as ONE mlp_chain launch: each branch reads its
        columns of the anchor rows and writes its slice of the concatenated embedding."""
        from . import fused
        box = box_3d if box_3d.dtype == torch.float32 and box_3d.is_contiguous() else box_3d.float().contiguous()
        rows = box.reshape(-1, box.shape[-1])
        n, ld = rows.shape
        branches = [(self.pos_fc, X), (self.size_fc, W), (self.yaw_fc, SIN_YAW), (self.vel_fc, VX)]
        widths = [fused.plan_of(seq).out_dim for seq, _ in branches]
        out = torch.empty(n, sum(widths), device=box.device, dtype=torch.float32)
        jobs, col = [], 0
        for (seq, xcol), wdt in zip(branches, widths):
            jobs.append(dict(plan=fused.plan_of(seq), x=(rows, ld, xcol), out=(out, out.shape[1], col)))
            col += wdt
        if n:
            fused.run_chains(jobs, n, box.device)
        return out.reshape(box_3d.shape[:-1] + (out.shape[1],))


@PLUGIN_LAYERS.register_module()
class SparseBox3DRefinementModule(BaseModule):
    """detection3d/blocks.py:77-154."""

    def __init__(self, embed_dims=256, output_dim=11, num_cls=10, normalize_yaw=False, refine_yaw=False,
                 with_cls_branch=True, with_quality_estimation=False):
        super().__init__()
        self.embed_dims = embed_dims
        self.output_dim = output_dim
        self.num_cls = num_cls
        self.normalize_yaw = normalize_yaw
        self.refine_yaw = refine_yaw
        self.refine_state = [X, Y, Z, W, L, H]
        if self.refine_yaw:
            self.refine_state += [SIN_YAW, COS_YAW]
        self.layers = nn.Sequential(*linear_relu_ln(embed_dims, 2, 2), Linear(self.embed_dims, self.output_dim),
                                    Scale([1.0] * self.output_dim))
        self.with_cls_branch = with_cls_branch
        if with_cls_branch:
            self.cls_layers = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(self.embed_dims, self.num_cls))
        self.with_quality_estimation = with_quality_estimation
        if with_quality_estimation:
            self.quality_layers = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(self.embed_dims, 2))

    def init_weight(self):
        if self.with_cls_branch:
            nn.init.constant_(self.cls_layers[-1].bias, bias_init_with_prob(0.01))

    def forward(self, instance_feature, anchor, anchor_embed, time_interval=1.0, return_cls=True, norm=None):
        """norm (an nn.LayerNorm, or None): the decoder's `norm` operator in front of this head has NOT been applied to
        instance_feature yet; the head's chain launch applies it (and leaves its output in self.norm_out), or, on the
        routes without that stage, it is applied here first."""
        fused_ok = instance_feature.is_cuda
        one_launch = fused_ok and not self.normalize_yaw and self.output_dim == 11 and anchor.shape[-1] == 11 and self.refine_yaw
        self.norm_out = None
        if norm is not None and not (one_launch and routes.R.chain_rows4):
            from . import dense
            instance_feature = self.norm_out = dense.layernorm(instance_feature, norm)
            norm = None
        if one_launch:
            # the whole head in one launch: MLP, Scale, then :133-143 (anchor added on every state, the
            # velocity columns divided by the time step first) as the chain's post stage
            from . import fused
            bs = instance_feature.shape[0]
            if not isinstance(time_interval, torch.Tensor):
                time_interval = instance_feature.new_tensor(time_interval)
            ti = time_interval.to(torch.float32).reshape(-1)
            ti = (ti.expand(bs) if ti.numel() == 1 else ti).contiguous()
            af, lda = fused._rows(anchor, anchor.shape[-1])
            post = dict(kind=fused.POST_REFINE3D, res=(af, lda), res_cols=11, div=ti,
                        div_rows=instance_feature.shape[1], div_col0=VX)
            xf, ldx = fused._rows(instance_feature, instance_feature.shape[-1])
            ef, lde = fused._rows(anchor_embed, anchor_embed.shape[-1])
            n, lead = xf.shape[0], instance_feature.shape[:-1]
            out = torch.empty(n, self.output_dim, device=xf.device)
            ln_w = ln_r = None
            if norm is not None:   # every chain normalises its rows itself; the first one writes the operator's output
                normed = torch.empty(n, xf.shape[1], device=xf.device)
                self.norm_out = normed.reshape(instance_feature.shape)
                ln_w, ln_r = (norm, (normed, xf.shape[1])), (norm, None)
            jobs = [dict(plan=fused.plan_of(self.layers), x=(xf, ldx, 0), x2=(ef, lde, 0), out=(out, self.output_dim, 0),
                         post=post, ln=ln_w)]
            cls = quality = None
            if return_cls:
                assert self.with_cls_branch, "Without classification layers !!!"
                cls = torch.empty(n, self.num_cls, device=xf.device)  # cls_layers(feature), :145-147
                jobs.append(dict(plan=fused.plan_of(self.cls_layers), x=(xf, ldx, 0), out=(cls, self.num_cls, 0), ln=ln_r))
                if self.with_quality_estimation:  # quality_layers(feature + embed), :149-152
                    quality = torch.empty(n, 2, device=xf.device)
                    jobs.append(dict(plan=fused.plan_of(self.quality_layers), x=(xf, ldx, 0), x2=(ef, lde, 0),
                                     out=(quality, 2, 0), ln=ln_r))
                    quality = quality.reshape(lead + (2,))
                cls = cls.reshape(lead + (self.num_cls,))
            if n:
                fused.run_chains(jobs, n, xf.device)
            return out.reshape(lead + (self.output_dim,)), cls, quality
        if fused_ok:
            from . import fused
            output = fused.chain_forward(self.layers, instance_feature, anchor_embed)
        else:
            feature = instance_feature + anchor_embed
            output = self.layers(feature)
        # :133 adds the anchor on the refined states; written as one add of a masked anchor so no
        # advanced-index scatter is launched (refine_state is a prefix 0..5 or 0..7)
        n_ref = len(self.refine_state)
        if not isinstance(time_interval, torch.Tensor):
            time_interval = instance_feature.new_tensor(time_interval)
        head = output[..., :n_ref] + anchor[..., :n_ref]
        parts = [head]
        if n_ref < VX:
            parts.append(output[..., n_ref:VX])
        if self.normalize_yaw:
            raise NotImplementedError("normalize_yaw is off in the SimPB configs")
        if self.output_dim > 8:
            ti = time_interval.reshape(-1, *([1] * (output.dim() - 1))) if time_interval.dim() else time_interval
            parts.append(output[..., VX:] / ti + anchor[..., VX:])  # :138-143
        output = torch.cat(parts, dim=-1)
        if return_cls:
            assert self.with_cls_branch, "Without classification layers !!!"
            if fused_ok:
                cls, quality = self._heads_fused(instance_feature, anchor_embed)
                return output, cls, quality
            cls = self.cls_layers(instance_feature)
        else:
            cls = None
        quality = self.quality_layers(feature) if return_cls and self.with_quality_estimation else None
        return output, cls, quality

    def _heads_fused(self, instance_feature, anchor_embed):
        """cls_layers(feature) and quality_layers(feature + embed) (:145-152) in one launch."""
        from . import fused
        xf, ldx = fused._rows(instance_feature, instance_feature.shape[-1])
        ef, lde = fused._rows(anchor_embed, anchor_embed.shape[-1])
        n = xf.shape[0]
        lead = instance_feature.shape[:-1]
        cls = torch.empty(n, self.num_cls, device=xf.device)
        jobs = [dict(plan=fused.plan_of(self.cls_layers), x=(xf, ldx, 0), out=(cls, self.num_cls, 0))]
        quality = None
        if self.with_quality_estimation:
            quality = torch.empty(n, 2, device=xf.device)
            jobs.append(dict(plan=fused.plan_of(self.quality_layers), x=(xf, ldx, 0), x2=(ef, lde, 0), out=(quality, 2, 0)))
        fused.run_chains(jobs, n, xf.device)
        return cls.reshape(lead + (self.num_cls,)), (quality.reshape(lead + (2,)) if quality is not None else None)


@PLUGIN_LAYERS.register_module()
class SparseBox3DKeyPointsGenerator(BaseModule):
    """detection3d/blocks.py:157-284 (no temporal key points: the SimPB configs never pass
    T_cur2temp_list)."""

    def __init__(self, embed_dims=256, num_learnable_pts=0, fix_scale=None):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_learnable_pts = num_learnable_pts
        if fix_scale is None:
            fix_scale = ((0.0, 0.0, 0.0),)
        self.fix_scale = nn.Parameter(torch.tensor(fix_scale, dtype=torch.float32), requires_grad=False)
        self.num_pts = len(self.fix_scale) + num_learnable_pts
        if num_learnable_pts > 0:
            self.learnable_fc = Linear(self.embed_dims, num_learnable_pts * 3)

    def init_weight(self):
        if self.num_learnable_pts > 0:
            nn.init.xavier_uniform_(self.learnable_fc.weight)
            nn.init.constant_(self.learnable_fc.bias, 0.0)

    def forward(self, anchor, instance_feature=None, T_cur2temp_list=None, cur_timestamp=None, temp_timestamps=None):
        if T_cur2temp_list is not None or temp_timestamps is not None:
            raise NotImplementedError("temporal key points are not used by the SimPB configs")
        bs, num_anchor = anchor.shape[:2]
        size = anchor[..., None, W:H + 1].exp()
        key_points = self.fix_scale * size
        if self.num_learnable_pts > 0 and instance_feature is not None:
            learnable_scale = (
                self.learnable_fc(instance_feature).reshape(bs, num_anchor, self.num_learnable_pts, 3).sigmoid() - 0.5)
            key_points = torch.cat([key_points, learnable_scale * size], dim=-2)
        cos, sin = anchor[..., None, COS_YAW], anchor[..., None, SIN_YAW]
        # rotation about z (:202-212) written out instead of building a 3x3 per anchor
        x = cos * key_points[..., 0] - sin * key_points[..., 1]
        y = sin * key_points[..., 0] + cos * key_points[..., 1]
        key_points = torch.stack([x, y, key_points[..., 2]], dim=-1)
        return key_points + anchor[..., None, X:Z + 1]

    @staticmethod
    def anchor_projection(anchor, T_src2dst_list, src_timestamp=None, dst_timestamps=None, time_intervals=None):
        """detection3d/blocks.py:248-280 including its acknowledged quirk (:271-278): the yaw pair
        is rotated as [cos, sin] and written back in that order into the [sin, cos] slots."""
        if anchor.is_cuda and anchor.dim() == 3 and anchor.shape[-1] == 11:
            return _anchor_projection_hip(anchor, T_src2dst_list, src_timestamp, dst_timestamps, time_intervals)
        dst_anchors = []
        for i in range(len(T_src2dst_list)):
            vel = anchor[..., VX:]
            vel_dim = vel.shape[-1]
            T_src2dst = torch.unsqueeze(T_src2dst_list[i].to(dtype=anchor.dtype), dim=1)
            center = anchor[..., X:Z + 1]
            if time_intervals is not None:
                time_interval = time_intervals[i]
            elif src_timestamp is not None and dst_timestamps is not None:
                time_interval = (src_timestamp - dst_timestamps[i]).to(dtype=vel.dtype)
            else:
                time_interval = None
            if time_interval is not None:
                center = center - vel * time_interval[:, None, None]
            center = torch.matmul(T_src2dst[..., :3, :3], center[..., None]).squeeze(dim=-1) + T_src2dst[..., :3, 3]
            size = anchor[..., W:H + 1]
            yaw = torch.matmul(T_src2dst[..., :2, :2], anchor[..., SIN_YAW:COS_YAW + 1].flip(-1)[..., None]).squeeze(-1)
            vel = torch.matmul(T_src2dst[..., :vel_dim, :vel_dim], vel[..., None]).squeeze(-1)
            dst_anchors.append(torch.cat([center, size, yaw, vel], dim=-1))
        return dst_anchors

    @staticmethod
    def distance(anchor):
        return torch.norm(anchor[..., :2], p=2, dim=-1)


def _anchor_projection_hip(anchor, T_src2dst_list, src_timestamp, dst_timestamps, time_intervals):
    """anchor_projection as one launch per transform (csrc/rowops.hip) instead of three batched 3x3
    matmuls through the vendor GEMM (27-44 us each on 600 anchors) and ten elementwise kernels."""
    from .. import _lib
    from .ops import _ptr, _stream
    bs, n, _ = anchor.shape
    src = anchor.contiguous().float()
    outs = []
    for i, T in enumerate(T_src2dst_list):
        if time_intervals is not None:
            dt = time_intervals[i]
        elif src_timestamp is not None and dst_timestamps is not None:
            dt = src_timestamp - dst_timestamps[i]
        else:
            dt = None
        if dt is not None:
            dt = dt.to(device=src.device, dtype=torch.float32).reshape(bs).contiguous()
        T = T.to(device=src.device, dtype=torch.float32).reshape(bs, 4, 4).contiguous()
        out = torch.empty_like(src)
        _lib.check(_lib.lib().simpb_anchor_projection(_ptr(out), _ptr(src), _ptr(T), _ptr(dt) if dt is not None else None,
                                                      bs, n, _stream()), "simpb_anchor_projection")
        outs.append(out)
    return outs


def bbox_cxcywh_to_xyxy(bbox):
    """mmdet.core.bbox.transforms.bbox_cxcywh_to_xyxy [mmdet 2.28.2, restated]."""
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


@BBOX_CODERS.register_module()
class SparseBox3DDecoder(object):
    """detection3d/decoder.py:9-252. Everything up to the final per-sample packing stays on the
    device; outputs are moved to the host once per field like the reference does (:230-251)."""

    def __init__(self, num_output=300, score_threshold=None, sorted=True):
        self.num_output = num_output
        self.score_threshold = score_threshold
        self.sorted = sorted

    def decode_box(self, box):
        yaw = torch.atan2(box[:, SIN_YAW], box[:, COS_YAW])
        return torch.cat([box[:, X:Z + 1], box[:, W:H + 1].exp(), yaw[:, None], box[:, VX:]], dim=-1)

    def decode_box2d(self, box, aug_config):
        crop = aug_config["crop"]
        scale_factor = aug_config["resize"]
        crop_img_size = (crop[2] - crop[0], crop[3] - crop[1])
        box = bbox_cxcywh_to_xyxy(box)
        box[..., 0::2] = (box[..., 0::2] * crop_img_size[0]).clamp(min=0, max=crop_img_size[0])
        box[..., 1::2] = (box[..., 1::2] * crop_img_size[1]).clamp(min=0, max=crop_img_size[1]) + crop[1]
        return box / scale_factor

    def _rank(self, cls_scores, qulity, output_idx, squeeze_cls):
        """The shared top-k / centerness re-score / sort prologue (:133-167 == :59-93)."""
        cls_scores = cls_scores[output_idx].sigmoid()
        cls_ids = None
        if squeeze_cls:
            cls_scores, cls_ids = cls_scores.max(dim=-1)
            cls_scores = cls_scores.unsqueeze(dim=-1)
        bs, num_pred, num_cls = cls_scores.shape
        flat = cls_scores.flatten(start_dim=1)
        on_gpu = flat.is_cuda and flat.shape[1] <= 2048 and self.sorted
        if on_gpu:  # one bitonic row sort (csrc/rowops.hip) instead of radix-select + radix-sort
            from .ops import topk_rows
            cls_scores, indices = topk_rows(flat, self.num_output)
        else:
            cls_scores, indices = flat.topk(self.num_output, dim=1, sorted=self.sorted)
        if not squeeze_cls:
            cls_ids = indices % num_cls
        mask = cls_scores >= self.score_threshold if self.score_threshold is not None else None
        cls_scores_origin = None
        if qulity is not None:
            centerness = torch.gather(qulity[output_idx][..., CNS], 1, indices // num_cls)
            cls_scores_origin = cls_scores.clone()
            cls_scores = cls_scores * centerness.sigmoid()
            if on_gpu:
                cls_scores, idx = topk_rows(cls_scores, cls_scores.shape[1])
            else:
                cls_scores, idx = torch.sort(cls_scores, dim=1, descending=True)
            if not squeeze_cls:
                cls_ids = torch.gather(cls_ids, 1, idx)
            if mask is not None:
                mask = torch.gather(mask, 1, idx)
            indices = torch.gather(indices, 1, idx)
        return cls_scores, cls_scores_origin, cls_ids, indices, mask, num_cls

    # ---- static-shape split of decode_with2d: everything with a fixed shape stays on the device
    # (and inside a captured frame); the variable-length 2D association is finished on the host
    # from two small fixed-shape records.
    def decode_static_device(self, cls_scores, box_preds, instance_id, qulity, cls_scores2d, box_preds2d, alloc,
                             aug_config, output_idx=-1, output_idx2d=-1):
        """Returns (rec3d f32 [bs, num_output, 15], rec2d f32 [bs, N2cap, 8]):
        rec3d = 10 decoded box + score + label + pre-centerness score + the int64 instance id bit-cast into two
        lanes (decoder.py:133-167, 23-34; include/simpb_hip.h);
        rec2d = 4 decoded box + score + label + rank of the slot's anchor in the sorted top-k (or -1)
        + camera of the slot (or -1)."""
        cls3, box3 = cls_scores[output_idx], box_preds[output_idx]
        if (routes.R.fused_decode and cls3.is_cuda and self.score_threshold is None and self.sorted and cls3.shape[1] <= 1024
                and self.num_output <= 512 and box3.shape[-1] == 11 and alloc.q2a.dtype == torch.int32):
            return self._decode_static_fused(cls3, box3, instance_id, qulity[output_idx] if qulity is not None else None,
                                             cls_scores2d[output_idx2d], box_preds2d[output_idx2d], alloc, aug_config)
        if getattr(alloc, "streams", 0):
            raise NotImplementedError("a batch of independent streams is decoded by the device kernels (csrc/decode.hip) only")
        scores, origin, cls_ids, indices, mask, num_cls = self._rank(cls_scores, qulity, output_idx, True)
        if mask is not None:
            raise NotImplementedError("score_threshold with the static decoder")
        box = torch.gather(box_preds[output_idx], 1, indices[..., None].expand(-1, -1, box_preds[output_idx].shape[-1]))
        bs, k, d = box.shape
        box3d = self.decode_box(box.reshape(bs * k, d)).reshape(bs, k, -1)
        labels = torch.gather(cls_ids, 1, indices)
        ids = torch.gather(instance_id, 1, indices)
        from ..dist import ids_to_lanes
        rec3d = torch.cat([box3d, scores[..., None], labels[..., None].to(box3d.dtype), origin[..., None],
                           ids_to_lanes(ids)], dim=-1)
        q2a = alloc.q2a.long()
        num_anchor = box_preds[output_idx].shape[1]
        rank_of_anchor = torch.full((bs, num_anchor + 1), -1, dtype=torch.long, device=q2a.device)
        rank_of_anchor.scatter_(1, indices, torch.arange(k, device=q2a.device)[None].expand(bs, -1))
        slot_rank = torch.gather(rank_of_anchor, 1, torch.where(q2a >= 0, q2a, num_anchor))
        s2d, l2d = cls_scores2d[output_idx2d].sigmoid().max(dim=-1)
        box2d = self.decode_box2d(box_preds2d[output_idx2d], aug_config)
        cam = alloc.query_cam[None].expand(bs, -1)
        rec2d = torch.cat([box2d, s2d[..., None], l2d[..., None].to(box2d.dtype), slot_rank[..., None].to(box2d.dtype),
                           cam[..., None].to(box2d.dtype)], dim=-1)
        return rec3d, rec2d

    def _decode_static_fused(self, cls3, box3, instance_id, quality, cls2d, box2d, alloc, aug_config):
        """The two records in two launches (csrc/decode.hip)."""
        from .. import _lib
        from .ops import _ptr, _stream
        lib = _lib.lib()
        bs, num_anchor, num_cls = cls3.shape
        k = self.num_output
        dev = cls3.device
        cls3, box3 = cls3.contiguous().float(), box3.contiguous().float()
        quality = quality.contiguous().float() if quality is not None else None
        ids = instance_id.contiguous().long() if instance_id is not None else None
        rec3d = torch.empty(bs, k, 15, device=dev)
        rank = torch.empty(bs, num_anchor, dtype=torch.int32, device=dev)
        _lib.check(lib.simpb_decode3d_record(_ptr(rec3d), _ptr(rank), _ptr(cls3), _ptr(quality) if quality is not None else None,
                                             _ptr(box3), _ptr(ids) if ids is not None else None, bs, num_anchor, num_cls, k,
                                             _stream()), "simpb_decode3d_record")
        cls2d, box2d = cls2d.contiguous().float(), box2d.contiguous().float()
        n2 = cls2d.shape[1]
        crop, resize = aug_config["crop"], aug_config["resize"]
        if getattr(alloc, "streams", 0):
            # independent streams: the flat slot array [1, bs * capacity] back to one record per stream, as a batch of one
            # writes it (its slots first, cameras counted within the stream, pad rows behind)
            if alloc.streams != bs or n2 % bs:
                raise ValueError("decode: the allocation does not belong to this batch")
            rows = n2 // bs
            rec2d = torch.empty(bs, rows, 8, device=dev)
            cams = (alloc.group_start.numel() - 1) // bs
            _lib.check(lib.simpb_decode2d_record_ragged(
                _ptr(rec2d), _ptr(cls2d), _ptr(box2d), _ptr(alloc.q2a.contiguous()), _ptr(alloc.query_cam.contiguous()),
                _ptr(alloc.group_start), _ptr(rank), bs, rows, cams, cls2d.shape[-1], num_anchor, float(crop[2] - crop[0]),
                float(crop[3] - crop[1]), float(crop[1]), float(resize), _stream()), "simpb_decode2d_record_ragged")
            return rec3d, rec2d
        rec2d = torch.empty(bs, n2, 8, device=dev)
        if n2:
            _lib.check(lib.simpb_decode2d_record(_ptr(rec2d), _ptr(cls2d), _ptr(box2d), _ptr(alloc.q2a.contiguous()),
                                                 _ptr(alloc.query_cam.contiguous()), _ptr(rank), bs, n2, cls2d.shape[-1],
                                                 num_anchor, float(crop[2] - crop[0]), float(crop[3] - crop[1]),
                                                 float(crop[1]), float(resize), _stream()), "simpb_decode2d_record")
        return rec3d, rec2d

    @staticmethod
    def decode_static_host(rec3d, rec2d, num_cams=6, independent=False):
        """Host half: the reference's per-sample dict (decoder.py:176-251) from the two records.
        Plain numpy on purpose: these are a few hundred elements, and CPU tensor ops would wake
        torch's intra-op thread pool once per call (measured: 50 ms stalls on a shared box).
        independent: the records are those of independent streams (SimPBHead.independent_streams): each is decoded as
        the batch of one it is, instead of with the group table of sample 0 carried along the batch (:216)."""
        import numpy as np
        rec3d, rec2d = np.asarray(rec3d), np.asarray(rec2d)
        if independent and len(rec3d) > 1:
            return [SparseBox3DDecoder.decode_static_host(rec3d[i:i + 1], rec2d[i:i + 1], num_cams)[0] for i in range(len(rec3d))]
        cam_all = rec2d[0][:, 7].astype(np.int64)
        query_groups, start = [], 0
        for c in range(num_cams):  # slots are camera-major: group c = the run of slots with camera c
            n = int((cam_all == c).sum())
            query_groups.append((start, start + n))
            start += n
        output = []
        for r3, r2 in zip(rec3d, rec2d):
            rank = r2[:, 6].astype(np.int64)
            idx2d = np.nonzero(rank >= 0)[0]
            trans_t = np.zeros((r3.shape[0], len(idx2d)), np.float32)
            trans_t[rank[idx2d], np.arange(len(idx2d))] = 1.0
            camidx_2d, query_groups_new = [], []
            for cam_idx, qg in enumerate(query_groups):
                part = np.nonzero((qg[0] <= idx2d) & (idx2d < qg[1]))[0]
                if len(part) > 0:
                    qg_new = (int(part[0]), int(part[-1]) + 1)
                elif len(query_groups_new) > 0:
                    qg_new = (query_groups_new[-1][-1], query_groups_new[-1][-1])
                else:
                    qg_new = (0, 0)
                camidx_2d.append(np.full(len(part), float(cam_idx), np.float32))
                query_groups_new.append(qg_new)
            query_groups = query_groups_new  # the reference re-binds the loop variable (:216)
            t = torch.from_numpy
            output.append({
                "boxes_3d": t(r3[:, :10].copy()), "scores_3d": t(r3[:, 10].copy()), "labels_3d": t(r3[:, 11].astype(np.int64)),
                "cls_scores": t(r3[:, 12].copy()), "instance_ids": t(np.ascontiguousarray(r3[:, 13:15]).view(np.int64)[:, 0].copy()),
                "boxes_2d": t(r2[idx2d, :4]), "scores_2d": t(r2[idx2d, 4]), "labels_2d": t(r2[idx2d, 5].astype(np.int64)),
                "camidx_2d": t(np.concatenate(camidx_2d)), "trans_matrix": t(trans_t), "query_groups": query_groups,
            })
        return output

    def decode(self, cls_scores, box_preds, instance_id=None, qulity=None, output_idx=-1):
        squeeze_cls = instance_id is not None
        cls_scores, origin, cls_ids, indices, mask, num_cls = self._rank(cls_scores, qulity, output_idx, squeeze_cls)
        box_preds = box_preds[output_idx]
        output = []
        for i in range(cls_scores.shape[0]):
            category_ids = cls_ids[i][indices[i]] if squeeze_cls else cls_ids[i]
            scores = cls_scores[i]
            box = box_preds[i, indices[i] // num_cls]
            if mask is not None:
                category_ids, scores, box = category_ids[mask[i]], scores[mask[i]], box[mask[i]]
            out = {"boxes_3d": self.decode_box(box).cpu(), "scores_3d": scores.cpu(), "labels_3d": category_ids.cpu()}
            if origin is not None:
                out["cls_scores"] = (origin[i][mask[i]] if mask is not None else origin[i]).cpu()
            if instance_id is not None:
                ids = instance_id[i, indices[i]]
                out["instance_ids"] = ids[mask[i]] if mask is not None else ids
            output.append(out)
        return output

    def decode_with2d(self, cls_scores, box_preds, instance_id=None, qulity=None, output_idx=-1, cls_scores2d=None,
                      box_preds2d=None, trans_matrix=None, query_groups=None, output_idx2d=-1, aug_configs=None,
                      with_association=False):
        """decoder.py:124-252. `trans_matrix[k]` may be the reference's dense one-hot
        [bs, N2, N3] tensor or an index table q2a i32[bs, N2] (slot -> anchor, -1 for pads)."""
        squeeze_cls = instance_id is not None
        cls_scores, origin, cls_ids, indices, mask, num_cls = self._rank(cls_scores, qulity, output_idx, squeeze_cls)
        box_preds = box_preds[output_idx]
        cls_scores2d = cls_scores2d[output_idx2d]
        box_preds2d = box_preds2d[output_idx2d]
        trans = trans_matrix[output_idx2d]
        query_groups = query_groups[output_idx2d]
        aug_config = aug_configs[0]
        num_anchor = box_preds.shape[1]
        output = []
        for i in range(cls_scores.shape[0]):
            category_ids = cls_ids[i][indices[i]] if squeeze_cls else cls_ids[i]
            scores = cls_scores[i]
            box = box_preds[i, indices[i] // num_cls]
            assert num_cls == 1
            if with_association:
                if trans.dtype in (torch.int32, torch.int64):  # index form: slot -> anchor
                    q2a = trans[i].long()
                    rank_of_anchor = torch.full((num_anchor + 1,), -1, dtype=torch.long, device=q2a.device)
                    rank_of_anchor[indices[i]] = torch.arange(indices.shape[1], device=q2a.device)
                    rank = rank_of_anchor[torch.where(q2a >= 0, q2a, num_anchor)]
                    indices2d = torch.where(rank >= 0)[0]
                    trans_t = torch.zeros(indices.shape[1], len(indices2d), device=q2a.device)
                    trans_t[rank[indices2d], torch.arange(len(indices2d), device=q2a.device)] = 1.0
                    trans_t = trans_t.cpu()
                else:
                    trans_t = trans[i].permute(1, 0)[indices[i]]
                    indices2d = torch.where(trans_t.any(0))[0]
                    trans_t = torch.index_select(trans_t, 1, indices2d).cpu()
            else:
                indices2d = torch.arange(len(box_preds2d[i]), device=box_preds2d.device)
                trans_t = None
            idx2d_host = indices2d.cpu()
            camidx_2d, query_groups_new = [], []
            for cam_idx, qg in enumerate(query_groups):
                parts_index = torch.where(torch.logical_and(qg[0] <= idx2d_host, idx2d_host < qg[1]))[0]
                if len(parts_index) > 0:
                    qg_new = (parts_index[0].item(), parts_index[-1].item() + 1)
                elif len(query_groups_new) > 0:
                    qg_new = (query_groups_new[-1][-1], query_groups_new[-1][-1])
                else:
                    qg_new = (0, 0)
                camidx_2d.append(torch.ones((len(parts_index))) * cam_idx)
                query_groups_new.append(qg_new)
            camidx_2d = torch.cat(camidx_2d, dim=0)
            query_groups = query_groups_new  # the reference re-binds the loop variable (:216)
            scores2d, category_ids2d = cls_scores2d[i, indices2d].sigmoid().max(dim=-1)
            box2d = self.decode_box2d(box_preds2d[i, indices2d], aug_config)
            if mask is not None:
                category_ids, scores, box = category_ids[mask[i]], scores[mask[i]], box[mask[i]]
            out = {
                "boxes_3d": self.decode_box(box).cpu(), "scores_3d": scores.cpu(), "labels_3d": category_ids.cpu(),
                "boxes_2d": box2d.cpu(), "scores_2d": scores2d.cpu(), "labels_2d": category_ids2d.cpu(),
                "camidx_2d": camidx_2d, "trans_matrix": trans_t, "query_groups": query_groups,
            }
            if origin is not None:
                out["cls_scores"] = (origin[i][mask[i]] if mask is not None else origin[i]).cpu()
            if instance_id is not None:
                ids = instance_id[i, indices[i]]
                out["instance_ids"] = ids[mask[i]] if mask is not None else ids
            output.append(out)
        return output
