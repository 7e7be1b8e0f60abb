"""Small building blocks shared by the plugin modules: the stand-ins for the handful of mmcv
classes the reference's models import (Linear, Scale, BaseModule, Sequential, build_* helpers,
MultiheadAttention) so that parameter names line up with released checkpoints
(SURVEY.md §8b "Checkpoint compatibility")."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import routes
from .registry import ATTENTION, NORM_LAYERS

Linear = nn.Linear
NORM_LAYERS.register_module("LN", module=nn.LayerNorm)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class Sequential(nn.Sequential):
    pass


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable per-element multiplier stored under `.scale`."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


def build_norm_layer(cfg, num_features):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    cfg.pop("requires_grad", None)
    if typ != "LN":
        raise KeyError(f"norm layer {typ} is not used on this path")
    return "ln", nn.LayerNorm(num_features, **cfg)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    return {"ReLU": nn.ReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid}[typ](**cfg)


def build_dropout(cfg):
    cfg = dict(cfg)
    if cfg.pop("type") != "Dropout":
        raise KeyError("only Dropout is used on this path")
    return nn.Dropout(p=cfg.get("drop_prob", 0.5), inplace=cfg.get("inplace", False))


def linear_relu_ln(embed_dims, in_loops, out_loops, input_dims=None):
    """models/blocks.py:32-43."""
    if input_dims is None:
        input_dims = embed_dims
    layers = []
    for _ in range(out_loops):
        for _ in range(in_loops):
            layers.append(Linear(input_dims, embed_dims))
            layers.append(nn.ReLU(inplace=True))
            input_dims = embed_dims
        layers.append(nn.LayerNorm(embed_dims))
    return layers


# Measured on MI355X (round 1): forking the value branch of each attention operator onto a side stream
# LOSES time inside the replayed frame graph (179-188 -> 163 frames/s): a cross-stream edge in a
# hipGraph costs more than two ~12 us GEMMs gain by overlapping. Kept for experiments (routes.parallel_branches), off.
_side_streams = {}


def run_parallel(fn_a, fn_b, device):
    """(fn_a(), fn_b()) with fn_b enqueued on a side HIP stream: two independent small GEMM chains
    (e.g. the q/k projection and the value path of an attention operator) each occupy well under
    half of the 256 CUs, so they are forked and joined instead of serialised. Works the same inside
    a graph capture (the fork/join becomes graph edges) and in eager mode."""
    if not routes.R.parallel_branches or device.type != "cuda":
        return fn_a(), fn_b()
    cur = torch.cuda.current_stream(device)
    key = (device.index, cur.cuda_stream)
    side = _side_streams.get(key)
    if side is None:
        side = _side_streams[key] = torch.cuda.Stream(device=device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = fn_b()
    a = fn_a()
    cur.wait_stream(side)
    for t in (b if isinstance(b, (tuple, list)) else (b,)):
        if torch.is_tensor(t):
            t.record_stream(cur)
    return a, b


def mha_forward(attn, query, key, value, groups=None, same_qk=False, query_cam=None, group_start=None, value_pre=None):
    """Arithmetic of torch.nn.MultiheadAttention.forward for batch-first [bs, N, E] inputs, using
    `attn` (an nn.MultiheadAttention) purely as the parameter container so checkpoint keys stay
    `attn.in_proj_weight/in_proj_bias/out_proj.*`.

    Camera groups: the reference builds an N x N additive mask that is 0 inside a block and -inf
    across (group_attn.py:104-113); softmax under that mask is exactly an independent softmax per
    block. On the GPU the attention core is csrc/attention.hip, which takes the group structure as
    device tables (query_cam, group_start) -- no mask tensor, no cross-group score work, static
    shapes. `groups` (a Python list of (start, end)) is the CPU/legacy form of the same thing."""
    e = attn.embed_dim
    h = attn.num_heads
    hd = e // h
    w, b = attn.in_proj_weight, attn.in_proj_bias
    bs, nq, _ = query.shape
    def qk_path():
        if same_qk:
            qk = F.linear(query, w[: 2 * e], b[: 2 * e])
            return qk[..., :e], qk[..., e:]
        return F.linear(query, w[:e], b[:e]), F.linear(key, w[e: 2 * e], b[e: 2 * e])

    def v_path():  # value_pre: a projection applied to the raw value first (fc_before, simpb_head.py:303)
        return F.linear(value_pre(value) if value_pre is not None else value, w[2 * e:], b[2 * e:])

    (q, k), v = run_parallel(qk_path, v_path, query.device)
    if q.is_cuda and hd == 64:
        from .ops import attention_f32
        if query_cam is not None and group_start is None:
            raise ValueError("grouped attention on the GPU needs the device group table (group_start)")
        o = attention_f32(q, k, v, h, query_cam, group_start)
        return F.linear(o, attn.out_proj.weight, attn.out_proj.bias)
    q = q.reshape(bs, nq, h, hd).transpose(1, 2)
    k = k.reshape(bs, -1, h, hd).transpose(1, 2)
    v = v.reshape(bs, -1, h, hd).transpose(1, 2)
    if groups is None or len(groups) <= 1:
        o = F.scaled_dot_product_attention(q, k, v)
    else:
        o = torch.zeros_like(q)  # rows outside every block are fully masked -> nan_to_num -> 0 (:131)
        for s, t in groups:
            if t > s:
                o[:, :, s:t] = F.scaled_dot_product_attention(q[:, :, s:t], k[:, :, s:t], v[:, :, s:t])
    o = o.transpose(1, 2).reshape(bs, nq, e)
    return F.linear(o, attn.out_proj.weight, attn.out_proj.bias)


def fused_graph_attention(layer, pre, post, query, query_pos, key=None, key_pos=None, value=None, query_cam=None,
                          group_start=None, m_live=None):
    """graph_model / graph_model2d of the reference (simpb_head.py:298-321) around one attention
    operator, as three launches: a grouped GEMM for the q/k/v projections reading `query | query_pos`
    (and `key | key_pos`) in place with fc_before folded into the value rows, the flash attention
    core, and one GEMM for post(identity + out_proj(o)) reading `o | query | query_pos`
    (plugin/dense.py). Returns None when the call is not one of the forms the decoder uses, and the
    caller takes the unfused route."""
    from . import dense
    from .ops import attention_f32
    attn = getattr(layer, "attn", None)
    if (not isinstance(attn, nn.MultiheadAttention) or not getattr(layer, "batch_first", False)
            or not isinstance(post, nn.Linear) or post.bias is not None or query_pos is None
            or query.shape != query_pos.shape or attn.in_proj_weight is None):
        return None
    e, h = attn.embed_dim, attn.num_heads
    if e != 2 * query.shape[-1] or e != h * 64 or query.shape[-1] % 64:
        return None
    if value is not None and (not isinstance(pre, nn.Linear) or pre.bias is not None):
        return None
    # routes.attention_split_fp16: the projections leave q / k / v as the half pairs the split-operand attention kernel
    # multiplies (softmax scale folded into the query rows), so that kernel converts nothing but its own probabilities
    halfs = routes.R.attention_split_fp16
    qs = 1.0 / math.sqrt(64.0) if halfs else None
    if key is None:
        if value is not None and value is not query:
            return None
        w, b = dense.fold_mha_in(attn, pre if value is not None else None, "qkv", q_scale=qs)
        qkv = dense.linear([query, query_pos], w, b, m_live=m_live, split_halfs=halfs)
        q, k, v = qkv[..., :e], qkv[..., e: 2 * e], qkv[..., 2 * e:]
    else:
        if key_pos is None or key.shape != key_pos.shape or (value is not None and value is not key) or query_cam is not None:
            return None
        wq, bq = dense.fold_mha_in(attn, None, "q", q_scale=qs)
        wkv, bkv = dense.fold_mha_in(attn, pre if value is not None else None, "kv")
        q, kv = dense.gemm(dense.job([query, query_pos], wq, bq, split_halfs=halfs),
                           dense.job([key, key_pos], wkv, bkv, split_halfs=halfs))
        k, v = kv[..., :e], kv[..., e:]
    o = attention_f32(q, k, v, h, query_cam, group_start, split=2 if halfs else 0)
    wo, bo = dense.fold_mha_out(attn, post)
    return dense.report(post, dense.linear([o, query, query_pos], wo, bo, m_live=m_live))


@ATTENTION.register_module()
class MultiheadAttention(BaseModule):
    """mmcv.cnn.bricks.transformer.MultiheadAttention as the reference configures it
    (config :167-173,200-215: batch_first=True, legacy `dropout=` kwarg). Eval only: dropouts
    are kept as modules for structure but are identity at inference."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer) if dropout_layer else None
        if "dropout" in kwargs:
            attn_drop = kwargs["dropout"]
            if dropout_layer is not None:
                dropout_layer["drop_prob"] = kwargs.pop("dropout")
            else:
                kwargs.pop("dropout")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, value_pre=None, **kwargs):
        if attn_mask is not None or key_padding_mask is not None:
            raise NotImplementedError("attention masks only occur on the reference's training path")
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
            key = query  # q and k share one input: their projections fuse into one GEMM
        else:
            same_qk = False
            if key_pos is not None:
                key = key + key_pos
        if not self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        out = mha_forward(self.attn, query, key, value, same_qk=same_qk, value_pre=value_pre)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))
