"""models/detection2d/blocks.py and models/utils.py of the reference: 2D sine/MLP query encoder
and the 2D refinement head."""
import math

import torch
import torch.nn as nn

from . import routes
from .layers import BaseModule, Linear, Scale, bias_init_with_prob, linear_relu_ln
from .registry import PLUGIN_LAYERS, POSITIONAL_ENCODING

__all__ = ["SparseBox2DRefinementModule", "SparseBox2DEncoder", "pos2posemb2d", "inverse_sigmoid"]


def inverse_sigmoid(x, eps=1e-5):
    """models/utils.py:4-8."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def pos2posemb2d(pos, num_pos_feats=128, temperature=10000):
    """models/utils.py:40-63: sine embedding, order cat(pos_y, pos_x[, pos_w, pos_h])."""
    scale = 2 * math.pi
    pos = pos * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=pos.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)

    def emb(col):
        p = pos[..., col, None] / dim_t
        return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)

    if pos.size(-1) == 2:
        return torch.cat((emb(1), emb(0)), dim=-1)
    if pos.size(-1) == 4:
        # :50-58 multiply width/height by 2*pi a second time; kept
        def emb2(col):
            p = (pos[..., col] * scale)[..., None] / dim_t
            return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)

        return torch.cat((emb(1), emb(0), emb2(2), emb2(3)), dim=-1)
    raise ValueError("Unknown pos_tensor shape(-1):{}".format(pos.size(-1)))


@POSITIONAL_ENCODING.register_module()
class SparseBox2DEncoder(BaseModule):
    """detection2d/blocks.py:20-63."""

    def __init__(self, embed_dims=256, with_size=False, with_sin_embed=False, mode="add", in_loops=1, out_loops=2):
        super().__init__()
        self.embed_dims = embed_dims
        self.mode = mode
        self.with_size = with_size
        self.with_sin_embed = with_sin_embed

        def embedding_layer(input_dims):
            return nn.Sequential(*linear_relu_ln(embed_dims, in_loops, out_loops, input_dims))

        if self.with_sin_embed:
            self.query_embeddings2d = embedding_layer(256)
        else:
            self.pos_fc = embedding_layer(2)
            if self.with_size:
                self.size_fc = embedding_layer(2)
                self.output_fc = embedding_layer(self.embed_dims)

    def forward(self, box_2d, m_live=None):
        if self.with_sin_embed:
            if box_2d.is_cuda and box_2d.shape[-1] == 2:
                from . import fused  # sine embedding + the two Linear/ReLU/LN stages in one launch
                return fused.chain_forward(self.query_embeddings2d, box_2d, sine=True, m_live=m_live)
            return self.query_embeddings2d(pos2posemb2d(box_2d))
        pos_feat = self.pos_fc(box_2d[..., :2])
        if not self.with_size:
            return pos_feat
        size_feat = self.size_fc(box_2d[..., 2:4])
        output = pos_feat + size_feat if self.mode == "add" else torch.cat([pos_feat, size_feat], dim=-1)
        return self.output_fc(output)


@PLUGIN_LAYERS.register_module()
class SparseBox2DRefinementModule(BaseModule):
    """detection2d/blocks.py:65-144 (depth branches are off in the SimPB configs)."""

    def __init__(self, embed_dims=256, output_dim=4, num_cls=10, alpha_dim=2, with_cls_branch=True,
                 with_alpha_branch=False, with_depth_branch=False, with_multibin_depth=False, depth_bin_num=64):
        super().__init__()
        self.embed_dims = embed_dims
        self.output_dim = output_dim
        self.num_cls = num_cls
        self.layers = nn.Sequential(*linear_relu_ln(embed_dims, 2, 2), Linear(self.embed_dims, self.output_dim),
                                    Scale([1.0] * self.output_dim))
        self.with_cls_branch = with_cls_branch
        if with_cls_branch:
            self.cls_layers = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(self.embed_dims, self.num_cls))
        self.with_alpha_branch = with_alpha_branch
        if with_alpha_branch:
            self.alpha_layers = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(self.embed_dims, alpha_dim),
                                              Scale([1.0] * 2))
        self.with_depth_branch = with_depth_branch
        self.with_multibin_depth = with_multibin_depth
        if with_depth_branch:
            raise NotImplementedError("the depth branch is off in the SimPB configs")

    def init_weight(self):
        if self.with_cls_branch:
            nn.init.constant_(self.cls_layers[-1].bias, bias_init_with_prob(0.01))

    def forward(self, instance_feature, anchor2d, anchor2d_embed, metas=None, return_cls=True, query_groups=None, m_live=None,
                norm=None):
        """norm: as SparseBox3DRefinementModule.forward (the `norm` operator in front of this head, not applied yet)."""
        fused_ok = instance_feature.is_cuda
        self.norm_out = None
        if norm is not None and not (fused_ok and routes.R.chain_rows4):
            from . import dense
            instance_feature = self.norm_out = dense.layernorm(instance_feature, norm, m_live=m_live)
            norm = None
        if fused_ok:
            from . import fused
            xf, ldx = fused._rows(instance_feature, instance_feature.shape[-1])
            ef, lde = fused._rows(anchor2d_embed, anchor2d_embed.shape[-1])
            n, lead = xf.shape[0], instance_feature.shape[:-1]
            out_t = torch.empty(n, self.output_dim, device=xf.device)
            # :122-125 + the final sigmoid (:144) as the chain's post stage
            af, lda = fused._rows(anchor2d, anchor2d.shape[-1])
            post = dict(kind=fused.POST_REFINE2D, res=(af, lda), res_cols=anchor2d.shape[-1])
            ln_w = ln_r = None
            if norm is not None:   # every chain normalises its rows itself; the first one writes the operator's output
                normed = torch.empty(n, xf.shape[1], device=xf.device)
                self.norm_out = normed.reshape(instance_feature.shape)
                ln_w, ln_r = (norm, (normed, xf.shape[1])), (norm, None)
            jobs = [dict(plan=fused.plan_of(self.layers), x=(xf, ldx, 0), x2=(ef, lde, 0), out=(out_t, self.output_dim, 0),
                         post=post, ln=ln_w)]
            cls_t = alpha_t = None
            if return_cls:
                cls_t = torch.empty(n, self.num_cls, device=xf.device)
                jobs.append(dict(plan=fused.plan_of(self.cls_layers), x=(xf, ldx, 0), out=(cls_t, self.num_cls, 0), ln=ln_r))
            if self.with_alpha_branch:
                adim = fused.plan_of(self.alpha_layers).out_dim
                alpha_t = torch.empty(n, adim, device=xf.device)
                jobs.append(dict(plan=fused.plan_of(self.alpha_layers), x=(xf, ldx, 0), out=(alpha_t, adim, 0), ln=ln_r))
            if n:
                fused.run_chains(jobs, n, xf.device, m_live=m_live)
            output = out_t.reshape(lead + (self.output_dim,))
        else:
            output = self.layers(instance_feature + anchor2d_embed)
        k = anchor2d.shape[-1]
        if k not in (2, 4):
            raise ValueError(k)
        if fused_ok:
            cls = cls_t.reshape(lead + (self.num_cls,)) if cls_t is not None else None
            alpha = alpha_t.reshape(lead + (alpha_t.shape[-1],)) if alpha_t is not None else None
            return output, cls, None, alpha
        output = torch.cat([output[..., :k] + inverse_sigmoid(anchor2d), output[..., k:]], dim=-1)  # :122-125
        cls = self.cls_layers(instance_feature) if return_cls else None
        alpha = self.alpha_layers(instance_feature) if self.with_alpha_branch else None
        return output.sigmoid(), cls, None, alpha
