"""Import shim that lets the reference's own model files (read from /root/reference, never
copied) run on CPU in the build container so golden vectors can be captured.

Three pieces, all outside /root/reference:
 1. namespace stand-ins for the un-vendored third-party packages mmcv-full==1.7.1 and
    mmdet==2.28.2 (requirement.txt:2-3). Anything arithmetic in here is restated from the
    packages' published semantics and is "parity unpinned" (no reference test pins it):
    MultiheadAttention wrapper, MultiScaleDeformableAttention parameters,
    multi_scale_deformable_attn_pytorch, Scale, bbox_cxcywh_to_xyxy.
 2. a CPU stand-in for the compiled deformable_aggregation op with the CUDA kernel's semantics
    (ops/src/deformable_aggregation_cuda.cu:129-187): bilinear, zero padding per tap,
    h_im = loc*H - 0.5, contribution dropped when loc is outside the open interval (0,1).
 3. group_attn.py is executed from its source text with the one device gate at :222 lifted so
    its own per-camera loop runs against the pure-PyTorch sampler.

This file only runs in the build container; the GPU box never sees /root/reference.
"""
import ast
import functools
import math
import sys
import types
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"
PLUGIN = REF_ROOT + "/projects/mmdet3d_plugin"


# ----------------------------------------------------------------------------- registries
class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **kw):
        return build_from_cfg(cfg, self, kw or None)


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


def _noop_decorator(*a, **k):
    def deco(fn):
        return fn

    return deco


# ----------------------------------------------------------------------------- mmcv pieces
class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


class Dropout(nn.Dropout):
    def __init__(self, drop_prob=0.5, inplace=False):
        super().__init__(p=drop_prob, inplace=inplace)


def build_dropout(cfg, default_args=None):
    cfg = dict(cfg)
    assert cfg.pop("type") == "Dropout"
    return Dropout(**cfg)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    return {"ReLU": nn.ReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid}[typ](**cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    cfg.pop("requires_grad", None)
    if typ == "LN":
        cfg.setdefault("eps", 1e-5)
        return "ln" + str(postfix), nn.LayerNorm(num_features, **cfg)
    raise KeyError(typ)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(p):
    return float(-math.log((1 - p) / p))


class MultiheadAttention(BaseModule):
    """mmcv.cnn.bricks.transformer.MultiheadAttention (mmcv-full 1.7.1), restated [mmcv-memory].
    The reference's own QueryGroupMultiheadAttention (group_attn.py:25-133) is a copy of it
    plus the group mask, which is what this restatement was checked against by reading."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=False, **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer)
        if "dropout" in kwargs:
            attn_drop = kwargs["dropout"]
            dropout_layer["drop_prob"] = kwargs.pop("dropout")
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None,
                attn_mask=None, key_padding_mask=None, **kwargs):
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
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class MultiScaleDeformableAttention(BaseModule):
    """Parameter container of mmcv's MultiScaleDeformableAttention [mmcv-memory]; the reference
    subclass overrides forward (group_attn.py:136-256) so only __init__ matters."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__(init_cfg)
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


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """mmcv.ops.multi_scale_deform_attn.multi_scale_deformable_attn_pytorch [mmcv-memory]; the
    CUDA op it mirrors is what group_attn.py:229-232 calls."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(h) * int(w) for h, w in value_spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (h, w) in enumerate(value_spatial_shapes):
        value_l = value_list[level].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, int(h), int(w))
        grid_l = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_list.append(
            F.grid_sample(value_l, grid_l, mode="bilinear", padding_mode="zeros", align_corners=False)
        )
    attention_weights = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries, num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1)
    return output.view(bs, num_heads * embed_dims, num_queries).transpose(1, 2).contiguous()


class MultiScaleDeformableAttnFunction:
    @staticmethod
    def apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step):
        return multi_scale_deformable_attn_pytorch(value, spatial_shapes, sampling_locations, attention_weights)


def bbox_cxcywh_to_xyxy(bbox):
    cx, cy, w, h = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def bbox_xyxy_to_cxcywh(bbox):
    x1, y1, x2, y2 = bbox.split((1, 1, 1, 1), dim=-1)
    return torch.cat([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


# ----------------------------------------------------------------------------- the CPU DAF
def daf_kernel_semantics(feat, spatial_shape, scale_start_index, loc, weights):
    """CPU stand-in for deformable_aggregation_forward with the CUDA kernel's semantics
    (deformable_aggregation_cuda.cu:13-59,129-187). feat [bs,N,C]; spatial_shape [cam,lvl,2];
    scale_start_index [cam,lvl]; loc [bs,A,P,cam,2]=(x,y) in image fraction;
    weights [bs,A,P,cam,lvl,G] -> [bs,A,C]."""
    bs, _, C = feat.shape
    num_cams, num_lvl = spatial_shape.shape[:2]
    _, A, P, _, _ = loc.shape
    G = weights.shape[-1]
    out = feat.new_zeros(bs, A, C)
    lx, ly = loc[..., 0], loc[..., 1]
    keep = (lx > 0) & (lx < 1) & (ly > 0) & (ly < 1)  # cu:169-171
    w_full = weights.repeat_interleave(C // G, dim=-1)  # channel c uses group c // (C/G), cu:149
    for cam in range(num_cams):
        for lvl in range(num_lvl):
            H, W = int(spatial_shape[cam, lvl, 0]), int(spatial_shape[cam, lvl, 1])
            start = int(scale_start_index[cam, lvl])
            fmap = feat[:, start:start + H * W].reshape(bs, H, W, C)
            h_im = (ly[:, :, :, cam] * H).double().sub(0.5).float()  # cu:180, float*int then -0.5 (double)
            w_im = (lx[:, :, :, cam] * W).double().sub(0.5).float()
            h_low, w_low = torch.floor(h_im), torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            val = feat.new_zeros(bs, A, P, C)
            bidx = torch.arange(bs)[:, None, None].expand(bs, A, P)
            for dy, dx, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
                yy, xx = h_low + dy, w_low + dx
                ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
                v = fmap[bidx, yy.clamp(0, H - 1), xx.clamp(0, W - 1)]
                val = val + (wt * ok)[..., None] * v
            contrib = val * w_full[:, :, :, cam, lvl] * keep[:, :, :, cam, None]
            out = out + contrib.sum(2)
    return out


# ----------------------------------------------------------------------------- install
_installed = {}


def install():
    """Create the stand-in packages and import the reference model files. Returns a namespace
    with the reference classes."""
    if _installed:
        return _installed["ns"]
    names = ["ATTENTION", "PLUGIN_LAYERS", "POSITIONAL_ENCODING", "FEEDFORWARD_NETWORK", "NORM_LAYERS",
             "TRANSFORMER_LAYER", "TRANSFORMER_LAYER_SEQUENCE"]
    regs = {n: Registry(n) for n in names}
    regs["NORM_LAYERS"].register_module("LN", module=nn.LayerNorm)
    regs["ATTENTION"].register_module("MultiheadAttention", module=MultiheadAttention)
    det = {n: Registry(n) for n in ["DETECTORS", "HEADS", "LOSSES", "BBOX_SAMPLERS", "BBOX_CODERS", "BBOX_ASSIGNERS",
                                    "BACKBONES", "NECKS"]}

    _mod("mmcv")
    _mod("mmcv.utils", build_from_cfg=build_from_cfg, Registry=Registry,
         deprecated_api_warning=_noop_decorator)
    _mod("mmcv.cnn", Linear=nn.Linear, Scale=Scale, bias_init_with_prob=bias_init_with_prob,
         build_activation_layer=build_activation_layer, build_norm_layer=build_norm_layer,
         xavier_init=xavier_init, constant_init=constant_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", **regs)
    _mod("mmcv.cnn.bricks.drop", build_dropout=build_dropout)

    class _Unused(BaseModule):
        def __init__(self, *a, **k):
            raise NotImplementedError("training/unused-by-config class")

    _mod("mmcv.cnn.bricks.transformer", FFN=_Unused, BaseTransformerLayer=_Unused,
         MultiScaleDeformableAttention=MultiScaleDeformableAttention, TransformerLayerSequence=_Unused,
         build_transformer_layer_sequence=None, MultiheadAttention=MultiheadAttention)
    _mod("mmcv.runner", BaseModule=BaseModule, force_fp32=_noop_decorator, auto_fp16=_noop_decorator)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule, Sequential=Sequential)
    _mod("mmcv.ops")
    _mod("mmcv.ops.multi_scale_deform_attn", MultiScaleDeformableAttnFunction=MultiScaleDeformableAttnFunction,
         multi_scale_deformable_attn_pytorch=multi_scale_deformable_attn_pytorch)

    _mod("mmdet")
    _mod("mmdet.core", reduce_mean=lambda x: x)
    _mod("mmdet.core.bbox")
    _mod("mmdet.core.bbox.builder", BBOX_SAMPLERS=det["BBOX_SAMPLERS"], BBOX_CODERS=det["BBOX_CODERS"],
         BBOX_ASSIGNERS=det["BBOX_ASSIGNERS"])
    _mod("mmdet.core.bbox.transforms", bbox_cxcywh_to_xyxy=bbox_cxcywh_to_xyxy,
         bbox_xyxy_to_cxcywh=bbox_xyxy_to_cxcywh)

    class HungarianAssigner:
        pass

    class AssignResult:
        pass

    _mod("mmdet.core.bbox.assigners", HungarianAssigner=HungarianAssigner)
    _mod("mmdet.core.bbox.assigners.assign_result", AssignResult=AssignResult)
    _mod("mmdet.core.bbox.match_costs", build_match_cost=lambda cfg: None)
    _mod("mmdet.models", DETECTORS=det["DETECTORS"], HEADS=det["HEADS"], LOSSES=det["LOSSES"],
         BaseDetector=BaseModule, build_backbone=None, build_head=None, build_neck=None)
    _mod("mmdet.models.builder", LOSSES=det["LOSSES"])

    # reference packages as bare namespace modules: submodules import unchanged, package
    # __init__ files (which pull datasets/apis and need nuscenes-devkit, cv2 ...) do not run.
    for name, path in [
        ("projects", REF_ROOT + "/projects"),
        ("projects.mmdet3d_plugin", PLUGIN),
        ("projects.mmdet3d_plugin.core", PLUGIN + "/core"),
        ("projects.mmdet3d_plugin.models", PLUGIN + "/models"),
        ("projects.mmdet3d_plugin.models.detection3d", PLUGIN + "/models/detection3d"),
        ("projects.mmdet3d_plugin.models.detection2d", PLUGIN + "/models/detection2d"),
    ]:
        m = _mod(name)
        m.__path__ = [path]
        m.__package__ = name

    import importlib

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        box3d = importlib.import_module("projects.mmdet3d_plugin.core.box3d")
        utils = importlib.import_module("projects.mmdet3d_plugin.models.utils")
        blocks = importlib.import_module("projects.mmdet3d_plugin.models.blocks")
        blocks.DAF = lambda feat, ss, ssi, loc, w: daf_kernel_semantics(
            feat.contiguous().float(), ss.int(), ssi.int(), loc.contiguous().float(), w.contiguous().float())
        d3_blocks = importlib.import_module("projects.mmdet3d_plugin.models.detection3d.blocks")
        d3_decoder = importlib.import_module("projects.mmdet3d_plugin.models.detection3d.decoder")
        d3_target = importlib.import_module("projects.mmdet3d_plugin.models.detection3d.target")
        d2_blocks = importlib.import_module("projects.mmdet3d_plugin.models.detection2d.blocks")
        d2_denoise = importlib.import_module("projects.mmdet3d_plugin.models.detection2d.denoise")
        bank = importlib.import_module("projects.mmdet3d_plugin.models.instance_bank")
        alloc = importlib.import_module("projects.mmdet3d_plugin.models.allocation")
        aggr = importlib.import_module("projects.mmdet3d_plugin.models.aggregation")

        # group_attn.py from its source text with the device gate at :222 lifted
        src = open(PLUGIN + "/models/group_attn.py").read()
        gate = "if torch.cuda.is_available() and value.is_cuda:"
        assert src.count(gate) == 1
        src = src.replace(gate, "if True:")
        ga = types.ModuleType("projects.mmdet3d_plugin.models.group_attn")
        ga.__package__ = "projects.mmdet3d_plugin.models"
        ga.__file__ = PLUGIN + "/models/group_attn.py"
        sys.modules[ga.__name__] = ga
        # QueryGroupDeformableDetrTransformerDecoder (:259-346) is unused by the config; its base
        # class only has to exist for the class statement to execute.
        sys.modules["mmcv.cnn.bricks.transformer"].TransformerLayerSequence = BaseModule
        exec(compile(src, ga.__file__, "exec"), ga.__dict__)

        head = importlib.import_module("projects.mmdet3d_plugin.models.simpb_head")

    # feature_maps_format from ops/__init__.py: the package cannot be imported (it needs the
    # compiled extension), so only that function's own source is compiled.
    tree = ast.parse(open(PLUGIN + "/ops/__init__.py").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "feature_maps_format"]
    ops_ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), PLUGIN + "/ops/__init__.py", "exec"), ops_ns)

    ns = types.SimpleNamespace(
        regs=regs, det=det, box3d=box3d, utils=utils, blocks=blocks, d3_blocks=d3_blocks, d3_decoder=d3_decoder,
        d3_target=d3_target, d2_blocks=d2_blocks, d2_denoise=d2_denoise, bank=bank, alloc=alloc, aggr=aggr,
        group_attn=ga, head=head, feature_maps_format=ops_ns["feature_maps_format"],
        build_from_cfg=build_from_cfg,
    )
    _installed["ns"] = ns
    return ns


def load_config(path=REF_ROOT + "/projects/configs/simpb_nus_r50_img_704x256.py"):
    scope = {}
    exec(compile(open(path).read(), path, "exec"), scope)
    return {k: v for k, v in scope.items() if not k.startswith("__")}


def head_cfg_for_eval(cfg, anchor):
    """config.model.head with the training-only pieces the stand-ins cannot build set to None
    and the k-means anchor file (not available offline) replaced by an array."""
    import copy

    h = copy.deepcopy(cfg["model"]["head"])
    h.pop("type")
    for k in list(h):
        if k.startswith("loss_") or k in ("coster2d", "coster3d"):
            h[k] = None
    h["instance_bank"]["anchor"] = anchor
    return h
