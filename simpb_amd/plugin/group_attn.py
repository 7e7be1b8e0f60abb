"""models/group_attn.py of the reference: camera-grouped self-attention and camera-grouped
multi-scale deformable cross-attention."""
import warnings

import torch
import torch.nn as nn

from . import dense
from .layers import BaseModule, build_dropout, mha_forward
from .ops import linear_f32, linear_split, ms_deform_attn_grouped, query_cam_from_groups

from . import routes
from .registry import ATTENTION


@ATTENTION.register_module()
class QueryGroupMultiheadAttention(BaseModule):
    """group_attn.py:25-133. The -inf block mask is never materialised: each camera group is an
    independent attention (see layers.mha_forward)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=False,
                 query_groups=None, **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer) if dropout_layer else None
        if "dropout" in kwargs:
            warnings.warn("The arguments `dropout` in MultiheadAttention has been deprecated", DeprecationWarning)
            attn_drop = kwargs["dropout"]
            dropout_layer["drop_prob"] = kwargs.pop("dropout")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.query_groups = query_groups
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.attn_mask = None
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, query_groups=None,
                group_attn_mask=None, key_padding_mask=None, query_cam=None, group_start=None, value_pre=None,
                **kwargs):
        if group_attn_mask is not None or key_padding_mask is not None:
            raise NotImplementedError("explicit masks are only used with with_allocate_attn_mask / training")
        same_qk = key is None
        if key is None:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if same_qk and key_pos is query_pos:
            key = query
        else:
            same_qk = False
            if key_pos is not None:
                key = key + key_pos
        if not self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        if query_groups is not None and self.query_groups != query_groups:
            self.query_groups = query_groups
        if query_cam is not None:  # device-side group table (what the allocation kernels emit)
            out = mha_forward(self.attn, query, key, value, same_qk=same_qk, query_cam=query_cam,
                              group_start=group_start, value_pre=value_pre)
        else:
            out = mha_forward(self.attn, query, key, value, groups=self.query_groups, same_qk=same_qk,
                              value_pre=value_pre)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


@ATTENTION.register_module()
class QueryGroupMultiScaleDeformableAttention(BaseModule):
    """group_attn.py:136-256 on top of mmcv's MultiScaleDeformableAttention parameters
    (sampling_offsets, attention_weights, value_proj, output_proj). The per-camera loop of
    :227-235 is one grouped kernel launch."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_cams=6, query_groups=None,
                 im2col_step=64, dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None, residual_mode="add"):
        super().__init__(init_cfg)
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        self.num_cams = num_cams
        self.query_groups = query_groups
        self.residual_mode = residual_mode
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def project_value(self, value, key_padding_mask=None):
        """value_proj over every camera token (:176-179): the largest GEMM of the decoder."""
        # fp32-grade product on the FP16 matrix cores (three split passes); the camera tokens are the output of
        # the fp16 backbone, well inside the half-precision range the split needs
        lin = linear_split if routes.R.split_value_proj else linear_f32
        half = getattr(value, "simpb_f16", None)   # the FPN left the same tokens in f16 (detector.FPN): two passes suffice
        if routes.R.split_value_proj and half is not None and half.shape == value.shape:
            value = half
        value = lin(value, self.value_proj.weight, self.value_proj.bias)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        return value

    def _forward_linear(self, query, query_pos, identity, value, value_f16, reference_points, spatial_shapes,
                        level_start_index, query_cam, m_live, keep_parts):
        """The operator with value_proj moved behind the sampling (csrc/msda_lin.hip): three launches -- offsets | logits
        product, sampling of the raw tokens (softmax and locations in its prologue), one product with the folded
        W_out . W_value -- and no value tensor at all."""
        from .ops import MSDA_LINEAR_WIDTH, msda_linear
        bs, nq, _ = query.shape
        w, b = dense.fold_stack("msda_in", [self.sampling_offsets, self.attention_weights], copies=2)
        both = dense.linear([query, query_pos], w, b, m_live=m_live)
        tokens = value_f16 if value_f16 is not None and value_f16.shape == value.shape else value
        # (a batch of independent streams arrives as ONE flat 2D set over all the streams' cameras: bs = 1 here and
        # value.shape[0] = streams x num_cams camera groups, allocation.allocate_independent)
        tokens = tokens.reshape(bs, value.shape[0] // bs, -1, self.embed_dims)
        agg = msda_linear(tokens.contiguous(), spatial_shapes, level_start_index, both, reference_points, query_cam, m_live)
        wf, bf = dense.fold_msda_linear(self.value_proj, self.output_proj, self.num_heads, MSDA_LINEAR_WIDTH)
        output = dense.report(self.output_proj, dense.linear(agg, wf, bf, m_live=m_live))
        if self.residual_mode == "add":
            return output + identity
        if self.residual_mode == "cat":
            output = dense.Segments([output, identity])
            return output if keep_parts else output.materialize()
        return output

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, query_cam=None,
                value_is_projected=False, m_live=None, keep_parts=False, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        fused = (routes.R.dense and query.is_cuda and query_pos is not None and self.batch_first
                 and query.shape == query_pos.shape)
        raw_query = query
        if query_pos is not None and not fused:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        linear_route = (routes.R.msda_linear and routes.R.dense and not value_is_projected and query.is_cuda and query_pos is not None
                        and self.batch_first and query.shape == query_pos.shape and key_padding_mask is None
                        and (self.num_heads, self.num_levels, self.num_points, self.embed_dims) == (8, 4, 4, 256)
                        and reference_points.shape[-1] == 2 and value.dim() == 3 and value.shape[0] % (bs * self.num_cams) == 0
                        and value.shape[-1] == 256 and query_cam is not None)
        if kwargs.get("ref_depth2d") is not None:   # group_attn.py:219-222 zeroes those locations; SimPBHead never passes it
            raise NotImplementedError("ref_depth2d masking is not used by SimPBHead")
        if linear_route:
            return self._forward_linear(raw_query, query_pos, identity, value, kwargs.get("value_f16"), reference_points,
                                        spatial_shapes, level_start_index, query_cam, m_live, keep_parts)
        if value_is_projected:  # project_value() was already applied by the caller (head.precompute_values)
            num_value = value.shape[-2] if value.dim() == 3 and value.shape[0] == bs * self.num_cams else value.numel() // (bs * self.num_cams * self.embed_dims)
        else:
            bcs, num_value, _ = value.shape
            assert bcs // self.num_cams == bs
            value = self.project_value(value, key_padding_mask)
        value = value.reshape(bs, self.num_cams, num_value, self.num_heads, -1)
        if reference_points.shape[-1] not in (2, 3):
            raise NotImplementedError("SimPB passes 2-d reference points (simpb_head.py:523)")
        if fused:
            # sampling_offsets(q + pos) and attention_weights(q + pos) as one product [q | pos] . [W | W]^T,
            # then softmax + reference point + offset / (W_l, H_l) in one launch (csrc/rowops.hip)
            from .. import _lib
            from .ops import _ptr, _stream
            w, b = dense.fold_stack("msda_in", [self.sampling_offsets, self.attention_weights], copies=2)
            both = dense.linear([raw_query, query_pos], w, b, m_live=m_live)
            ref, _, ldref = dense.rows2d(reference_points[..., :2])
            lp = self.num_levels * self.num_points
            sampling_locations = torch.empty(bs, num_query, self.num_heads, self.num_levels, self.num_points, 2,
                                             device=query.device)
            attention_weights = torch.empty(bs, num_query, self.num_heads, self.num_levels, self.num_points,
                                            device=query.device)
            shapes = spatial_shapes.contiguous().long()
            if both.shape[-1] != 3 * self.num_heads * lp:
                raise ValueError("sampling_offsets / attention_weights widths do not match heads x levels x points")
            _lib.check(_lib.lib().simpb_msda_prep(
                _ptr(sampling_locations), _ptr(attention_weights), _ptr(both), both.shape[-1], _ptr(ref), ldref,
                _ptr(shapes), bs * num_query, self.num_heads, self.num_levels, self.num_points,
                _ptr(m_live) if m_live is not None else None, _stream()), "simpb_msda_prep")
        else:
            sampling_offsets = self.sampling_offsets(query).view(
                bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
            attention_weights = self.attention_weights(query).view(
                bs, num_query, self.num_heads, self.num_levels * self.num_points).softmax(-1)
            attention_weights = attention_weights.view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        if kwargs.get("query_groups", None) is not None:
            self.query_groups = kwargs["query_groups"]
        if query_cam is None:
            query_cam = query_cam_from_groups(self.query_groups, num_query, query.device)
        output = ms_deform_attn_grouped(value, spatial_shapes, level_start_index, sampling_locations,
                                        attention_weights, query_cam)
        if fused:
            output = dense.linear(output, self.output_proj.weight, self.output_proj.bias, m_live=m_live)
        else:
            output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        output = self.dropout(output)
        if self.residual_mode == "add":
            output = output + identity
        elif self.residual_mode == "cat":
            output = dense.Segments([output, identity])
            if not keep_parts:
                output = output.materialize()
        return output
