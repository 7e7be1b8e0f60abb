"""models/blocks.py of the reference: DeformableFeatureAggregation, AsymmetricFFN (+ an inert
DenseDepthNet, which is an auxiliary training loss only: simpb.py:83-86,104-107)."""
import torch
import torch.nn as nn

from . import dense, routes
from .layers import (BaseModule, Linear, Sequential, build_activation_layer, build_dropout, build_norm_layer,
                     linear_relu_ln)
from .ops import deformable_aggregation_function as DAF
from .ops import dfa_fused
from .registry import ATTENTION, FEEDFORWARD_NETWORK, PLUGIN_LAYERS, build_from_cfg

__all__ = ["DeformableFeatureAggregation", "DenseDepthNet", "AsymmetricFFN"]


@ATTENTION.register_module()
class DeformableFeatureAggregation(BaseModule):
    """blocks.py:46-196. Only the compiled-operator branch exists here (the shipped config sets
    use_deformable_func=True, config :55,145); use_deformable_func=False is refused rather than
    silently routed to a PyTorch sampler whose border rule differs."""

    def __init__(self, embed_dims=256, num_groups=8, num_levels=4, num_cams=6, proj_drop=0.0, attn_drop=0.0,
                 kps_generator=None, temporal_fusion_module=None, use_temporal_anchor_embed=True,
                 use_deformable_func=False, use_camera_embed=False, residual_mode="add"):
        super().__init__()
        if embed_dims % num_groups != 0:
            raise ValueError(f"embed_dims must be divisible by num_groups, but got {embed_dims} and {num_groups}")
        if not use_deformable_func:
            raise NotImplementedError("only use_deformable_func=True (the HIP operator) is provided")
        if temporal_fusion_module is not None:
            raise NotImplementedError("temporal_fusion_module is not used by the SimPB configs")
        self.group_dims = embed_dims // num_groups
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_groups = num_groups
        self.num_cams = num_cams
        self.use_temporal_anchor_embed = use_temporal_anchor_embed
        self.use_deformable_func = use_deformable_func
        self.attn_drop = attn_drop
        self.residual_mode = residual_mode
        self.proj_drop = nn.Dropout(proj_drop)
        kps_generator = dict(kps_generator)
        kps_generator["embed_dims"] = embed_dims
        self.kps_generator = build_from_cfg(kps_generator, PLUGIN_LAYERS)
        self.num_pts = self.kps_generator.num_pts
        self.temp_module = None
        self.output_proj = Linear(embed_dims, embed_dims)
        if use_camera_embed:
            self.camera_encoder = Sequential(*linear_relu_ln(embed_dims, 1, 2, 12))
            self.weights_fc = Linear(embed_dims, num_groups * num_levels * self.num_pts)
        else:
            self.camera_encoder = None
            self.weights_fc = Linear(embed_dims, num_groups * num_cams * num_levels * self.num_pts)

    def init_weight(self):
        nn.init.constant_(self.weights_fc.weight, 0.0)
        nn.init.constant_(self.weights_fc.bias, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.0)

    def forward(self, instance_feature, anchor, anchor_embed, feature_maps, metas, keep_parts=False, cam_embed=None,
                **kwargs):
        if (instance_feature.is_cuda and self.camera_encoder is not None and metas.get("image_wh") is not None
                and getattr(self.kps_generator, "num_learnable_pts", 0) > 0):
            return self._forward_fused(instance_feature, anchor, anchor_embed, feature_maps, metas, keep_parts,
                                       cam_embed)
        bs, num_anchor = instance_feature.shape[:2]
        key_points = self.kps_generator(anchor, instance_feature)
        weights = self._get_weights(instance_feature, anchor_embed, metas)
        points_2d = (
            self.project_points(key_points, metas["projection_mat"], metas.get("image_wh"))
            .permute(0, 2, 3, 1, 4)
            .reshape(bs, num_anchor, self.num_pts, self.num_cams, 2)
        )
        weights = weights.permute(0, 1, 4, 2, 3, 5).contiguous().reshape(
            bs, num_anchor, self.num_pts, self.num_cams, self.num_levels, self.num_groups)
        features = DAF(*feature_maps, points_2d, weights).reshape(bs, num_anchor, self.embed_dims)
        output = self.proj_drop(self.output_proj(features))
        if self.residual_mode == "add":
            output = output + instance_feature
        elif self.residual_mode == "cat":
            output = torch.cat([output, instance_feature], dim=-1)
        return output

    def _forward_fused(self, instance_feature, anchor, anchor_embed, feature_maps, metas, keep_parts=False,
                       cam_embed=None):
        """Same dataflow as forward(), with the operand producers as two HIP kernels writing the
        aggregation kernel's own layouts (csrc/dfa_prep.hip) instead of ~25 PyTorch kernels, and the
        three Linear layers in front of them (learnable_fc on the feature, weights_fc on feature +
        anchor_embed and on the camera embedding) as one grouped GEMM launch."""
        from .. import _lib
        from . import fused
        from .ops import _ptr, _stream
        lib = _lib.lib()
        bs, num_anchor = instance_feature.shape[:2]
        kps = self.kps_generator
        dev = instance_feature.device
        anchor_c = anchor.contiguous().float()
        proj = metas["projection_mat"].contiguous().float()
        wh = metas["image_wh"].contiguous().float()
        if cam_embed is None:  # the head hands in the embeddings of all its layers, computed in one launch
            cam_embed = fused.chain_forward(self.camera_encoder, proj[:, :, :3].reshape(bs, self.num_cams, -1))
        # weights_fc(f + e) = [f | e] . [W | W]^T + b; weights_fc(f + e + c) = that + c . W^T (no second bias)
        if routes.R.dense:
            learn, feat_logits, cam_logits = dense.gemm(
                dense.job(instance_feature, kps.learnable_fc.weight, kps.learnable_fc.bias),
                dense.job([instance_feature, anchor_embed], dense.fold_sum_input(self.weights_fc), self.weights_fc.bias),
                dense.job(cam_embed, self.weights_fc.weight))
        else:
            from .ops import linear_f32
            learn = kps.learnable_fc(instance_feature).contiguous()
            feat_logits = linear_f32(instance_feature + anchor_embed, self.weights_fc.weight, self.weights_fc.bias)
            cam_logits = linear_f32(cam_embed, self.weights_fc.weight)
        shipped_layout = (self.num_cams == 6 and self.num_levels == 4 and self.num_groups == 8 and self.embed_dims == 256
                          and kps.fix_scale.shape[0] == 7 and kps.num_learnable_pts == 6)   # what the one-launch kernel is compiled for
        if routes.R.fused_dfa and routes.R.dense and shipped_layout:
            # key points + projection + weight softmax inside the aggregation launch (csrc/deform_agg_fused.hip), reading the
            # f16 copy of the tokens where the FPN left one (the tokens are f16 numbers: same bits, half the gather bytes)
            feat = feature_maps[0]
            half = getattr(feat, "simpb_f16", None) if routes.R.dfa_f16_tokens else None
            if half is not None and half.shape == feat.shape:
                feat = half
            features = dfa_fused(feat, feature_maps[1], feature_maps[2], anchor_c, learn, kps.fix_scale, proj, wh, feat_logits,
                                 cam_logits, self.num_groups)
            return self._project_out(features, instance_feature, keep_parts)
        num_fix = kps.fix_scale.shape[0]
        loc = torch.empty(bs, num_anchor, self.num_pts, self.num_cams, 2, device=dev)
        _lib.check(lib.simpb_dfa_points(_ptr(loc), None, _ptr(anchor_c), _ptr(learn), _ptr(kps.fix_scale), _ptr(proj),
                                        _ptr(wh), bs, num_anchor, num_fix, kps.num_learnable_pts, self.num_cams,
                                        _stream()), "simpb_dfa_points")
        weights = torch.empty(bs, num_anchor, self.num_pts, self.num_cams, self.num_levels, self.num_groups, device=dev)
        _lib.check(lib.simpb_dfa_weights(_ptr(weights), _ptr(feat_logits), _ptr(cam_logits), bs, num_anchor,
                                         self.num_cams, self.num_levels, self.num_pts, self.num_groups, _stream()),
                   "simpb_dfa_weights")
        features = DAF(*feature_maps, loc, weights).reshape(bs, num_anchor, self.embed_dims)
        return self._project_out(features, instance_feature, keep_parts)

    def _project_out(self, features, instance_feature, keep_parts):
        if routes.R.dense:
            output = dense.linear(features, self.output_proj.weight, self.output_proj.bias)
        else:
            output = self.output_proj(features)
        if self.residual_mode == "add":
            output = output + instance_feature
        elif self.residual_mode == "cat":
            output = dense.Segments([output, instance_feature])
            if not keep_parts:
                output = output.materialize()
        return output

    def _get_weights(self, instance_feature, anchor_embed, metas=None):
        """blocks.py:164-196 (eval: no attention dropout)."""
        bs, num_anchor = instance_feature.shape[:2]
        feature = instance_feature + anchor_embed
        if self.camera_encoder is not None:
            cam_in = metas["projection_mat"][:, :, :3].reshape(bs, self.num_cams, -1)
            if cam_in.is_cuda:
                from . import fused
                from .ops import linear_f32
                camera_embed = fused.chain_forward(self.camera_encoder, cam_in)
                # weights_fc is linear, so weights_fc(feature[:, :, None] + camera_embed[:, None]) =
                # weights_fc(feature)[:, :, None] + camera_embed @ W^T: one [N, 256] x [256, 416] product
                # plus a [cams, 256] one instead of the reference's [N * cams, 256] product (:177-179)
                logits = (linear_f32(feature, self.weights_fc.weight, self.weights_fc.bias)[:, :, None]
                          + linear_f32(camera_embed, self.weights_fc.weight)[:, None])
            else:
                camera_embed = self.camera_encoder(cam_in)
                logits = self.weights_fc(feature[:, :, None] + camera_embed[:, None])
        else:
            logits = self.weights_fc(feature)
        weights = (
            logits
            .reshape(bs, num_anchor, -1, self.num_groups)
            .softmax(dim=-2)
            .reshape(bs, num_anchor, self.num_cams, self.num_levels, self.num_pts, self.num_groups)
        )
        if self.training and self.attn_drop > 0:
            raise NotImplementedError("training-time attention dropout")
        return weights

    @staticmethod
    def project_points(key_points, projection_mat, image_wh=None):
        """blocks.py:198-213."""
        pts_extend = torch.cat([key_points, torch.ones_like(key_points[..., :1])], dim=-1)
        points_2d = torch.matmul(projection_mat[:, :, None, None], pts_extend[:, None, ..., None]).squeeze(-1)
        points_2d = points_2d[..., :2] / torch.clamp(points_2d[..., 2:3], min=1e-5)
        if image_wh is not None:
            points_2d = points_2d / image_wh[:, :, None, None]
        return points_2d


@PLUGIN_LAYERS.register_module()
class DenseDepthNet(BaseModule):
    """blocks.py:264-327 is an auxiliary depth loss used in training only. Registered so the
    config's `depth_branch` entry builds and released checkpoints load (same conv names)."""

    def __init__(self, embed_dims=256, num_depth_layers=1, equal_focal=100, max_depth=60, loss_weight=1.0):
        super().__init__()
        self.embed_dims = embed_dims
        self.equal_focal = equal_focal
        self.num_depth_layers = num_depth_layers
        self.max_depth = max_depth
        self.loss_weight = loss_weight
        self.depth_layers = nn.ModuleList(
            [nn.Conv2d(embed_dims, 1, kernel_size=1, stride=1, padding=0) for _ in range(num_depth_layers)])

    def forward(self, feature_maps, focal=None, gt_depths=None):
        raise NotImplementedError("DenseDepthNet is training-only (simpb.py:83-86)")


@FEEDFORWARD_NETWORK.register_module()
class AsymmetricFFN(BaseModule):
    """blocks.py:330-393."""

    def __init__(self, in_channels=None, pre_norm=None, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0, dropout_layer=None, add_identity=True,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        assert num_fcs >= 2, f"num_fcs should be no less than 2. got {num_fcs}."
        self.in_channels = in_channels
        self.pre_norm = pre_norm
        self.embed_dims = embed_dims
        self.feedforward_channels = feedforward_channels
        self.num_fcs = num_fcs
        self.act_cfg = act_cfg
        self.activate = build_activation_layer(act_cfg)
        layers = []
        if in_channels is None:
            in_channels = embed_dims
        if pre_norm is not None:
            self.pre_norm = build_norm_layer(pre_norm, in_channels)[1]
        for _ in range(num_fcs - 1):
            layers.append(Sequential(Linear(in_channels, feedforward_channels), self.activate, nn.Dropout(ffn_drop)))
            in_channels = feedforward_channels
        layers.append(Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = Sequential(*layers)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()
        self.add_identity = add_identity
        if self.add_identity:
            self.identity_fc = nn.Identity() if in_channels == embed_dims else Linear(self.in_channels, embed_dims)

    def _fusable(self):
        return (isinstance(self.pre_norm, nn.LayerNorm) and self.num_fcs == 2 and self.add_identity
                and isinstance(self.identity_fc, nn.Linear) and isinstance(self.activate, nn.ReLU)
                and isinstance(self.dropout_layer, (nn.Identity, nn.Dropout)))

    def forward(self, x, identity=None, m_live=None):
        first = x[0] if isinstance(x, dense.Segments) else x
        if routes.R.dense and first.is_cuda and identity is None and not self.training and self._fusable():
            # blocks.py:384-393 in three launches: pre-norm over the (possibly two-segment) input,
            # fc1 + ReLU, then [h | x] . [W_2 | W_id]^T + b_2 + b_id (dense.fold_ffn_out)
            fc1, fc2 = self.layers[0][0], self.layers[1]
            xn = dense.layernorm(x, self.pre_norm, m_live=m_live)
            h = dense.linear(xn, fc1.weight, fc1.bias, relu=True, m_live=m_live)
            w, b = dense.fold_ffn_out(fc2, self.identity_fc)
            return dense.linear([h, xn], w, b, m_live=m_live)
        if isinstance(x, dense.Segments):
            x = x.materialize()
        if self.pre_norm is not None:
            x = self.pre_norm(x)
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        if identity is None:
            identity = x
        identity = self.identity_fc(identity)
        return identity + self.dropout_layer(out)
