"""models/simpb.py of the reference (the SimPB detector) plus the two mmdet 2.28.2 classes it
builds and which are not vendored there: ResNet (style='pytorch') and FPN. Parameter names follow
mmdet so released checkpoints load: img_backbone.{conv1,bn1,layer{1-4}.{i}.{conv,bn}{1-3},
downsample.{0,1}}, img_neck.{lateral_convs,fpn_convs}.{i}.conv.* [mmdet, restated].

This is SURVEY.md §8(f) item 1 ("next": caller of the hot path): it runs on PyTorch-ROCm/MIOpen
in fp16 + channels_last, and hands the head a channel-last token buffer without the 92 MB
permute the reference pays in feature_maps_format (ops/__init__.py:78)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import BaseModule
from .ops import feature_maps_format, format_tokens
from .registry import BACKBONES, DETECTORS, HEADS, NECKS, PLUGIN_LAYERS, build_from_cfg

__all__ = ["SimPB", "ResNet", "FPN"]


# routes.conv1x1_kernel: conv1 / conv3 / downsample of the fp16 bottlenecks run as csrc/conv1x1.hip (one launch each,
# epilogue included); off: vendor convolution + csrc/bias_act.hip (two launches), also the cross-check in tests.
# routes.conv3x3_kernel: conv2 of the fp16 bottlenecks and the FPN's output convolutions run as csrc/conv3x3.hip (implicit
# GEMM, epilogue included; the FPN's write the decoder's tokens themselves); off: vendor convolution (+ csrc/bias_act.hip
# / the token format pass), also the cross-check in tests.
# routes.stem_epilogue_kernel: the stem's bias + ReLU + 3x3/2 max-pool as one pass (csrc/bias_act.hip); off: bias_act +
# PyTorch's pooling.
from . import routes

# A convolution whose shape misses an in-tree kernel's rules goes to the vendor library. That is legitimate for a module
# used on its own, but not silently: the vendor's solver for some shapes is built on the gfx950 double-K matrix
# instructions, which disturb the vector arithmetic of kernels running BESIDE them ("the two-stream
# fault") -- so the runners that overlap the backbone with a decoder (runner.PipelinedRunner) set STRICT_NO_VENDOR and
# the fallback raises there; elsewhere it warns once per site. A route switched off on purpose (routes.override: the
# tests' vendor cross-check) is the caller's decision and does neither.
STRICT_NO_VENDOR = False
_warned = set()


def _vendor_fallback(site, wanted, detail):
    if not wanted:
        return
    msg = (f"simpb_amd: {site} falls back to the vendor convolution ({detail}); the in-tree kernel's shape rules "
           "(input channels % 64, output channels % 8, channels_last fp16, stride 1 or 2) are not met")
    if STRICT_NO_VENDOR:
        raise RuntimeError(msg + " -- refused under a pipelined runner: a vendor kernel may issue the gfx950 double-K matrix "
                                 "instructions beside the decoder's kernels (DESIGN.md section 4)")
    if site not in _warned:
        import warnings
        _warned.add(site)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if getattr(self, "fused_epilogue", False) and x.is_cuda and x.dtype == torch.float16:
            return self._forward_fused(x)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)

    def _forward_fused(self, x):
        """BN already folded (SimPB.fuse_conv_bn): convolutions run without bias and each is followed
        by ONE epilogue kernel (bias [+ residual] [+ ReLU]) instead of add_, add and relu_."""
        from .ops import bias_act_, conv1x1_nhwc, conv3x3_nhwc

        def conv(m, t):
            return F.conv2d(t, m.weight, None, m.stride, m.padding)

        def takes(m, t):
            return (routes.R.conv1x1_kernel and m.in_channels % 64 == 0 and m.out_channels % 8 == 0 and m.stride[0] in (1, 2)
                    and m.stride[0] == m.stride[1] and t.is_contiguous(memory_format=torch.channels_last))

        def pointwise(m, t, residual=None, relu=True, input_bias=None):
            # the 1x1 convolutions with their whole epilogue in one launch (csrc/conv1x1.hip); anything the kernel
            # does not take (odd channel counts, an input that is not channels_last) goes the two-launch way
            if takes(m, t):
                return conv1x1_nhwc(t, m.weight, m.bias, residual, relu, m.stride[0], input_bias=input_bias)
            assert input_bias is None
            _vendor_fallback("bottleneck 1x1 convolution", routes.R.conv1x1_kernel, f"{m.in_channels}->{m.out_channels}, stride {m.stride}")
            return bias_act_(conv(m, t), m.bias, residual, relu=relu)

        identity = x if self.downsample is None else pointwise(self.downsample[0], x, None, relu=False)
        out = pointwise(self.conv1, x)
        c2 = self.conv2
        if (routes.R.conv3x3_kernel and c2.in_channels % 64 == 0 and c2.out_channels % 8 == 0 and c2.stride[0] in (1, 2)
                and c2.stride[0] == c2.stride[1] and c2.padding == (1, 1) and out.is_contiguous(memory_format=torch.channels_last)):
            # the 3x3 convolution as one implicit-GEMM launch with its bias + ReLU (csrc/conv3x3.hip)
            out = conv3x3_nhwc(out, c2.weight, c2.bias, relu=True, stride=c2.stride[0])
            return pointwise(self.conv3, out, identity, relu=True)
        _vendor_fallback("bottleneck 3x3 convolution", routes.R.conv3x3_kernel, f"{c2.in_channels}->{c2.out_channels}, stride {c2.stride}")
        out = conv(c2, out)
        if takes(self.conv3, out) and self.conv3.stride[0] == 1:
            # conv2's bias + ReLU are applied by conv3 while it stages its input: no epilogue pass for the 3x3 convolution
            return pointwise(self.conv3, out, identity, relu=True, input_bias=self.conv2.bias)
        out = bias_act_(out, self.conv2.bias, None, relu=True)
        return pointwise(self.conv3, out, identity, relu=True)


@BACKBONES.register_module()
class ResNet(BaseModule):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 style="pytorch", frozen_stages=-1, norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True,
                 with_cp=False, pretrained=None, init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if depth not in self.arch_settings or style != "pytorch":
            raise NotImplementedError("ResNet-50/101/152 with style='pytorch' (the SimPB configs)")
        self.depth = depth
        self.out_indices = out_indices
        self.pretrained = pretrained
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, blocks in enumerate(self.arch_settings[depth][:num_stages]):
            planes = 64 * 2 ** i
            layers = []
            for j in range(blocks):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * 4):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                         nn.BatchNorm2d(planes * 4))
                layers.append(Bottleneck(inplanes, planes, stride, down))
                inplanes = planes * 4
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)

    def _stem_kernel_ok(self, x):
        c1, mp = self.conv1, self.maxpool
        return (routes.R.stem_kernel and getattr(self, "fused_epilogue", False) and x.is_cuda and x.dtype == torch.float32
                and tuple(c1.weight.shape) == (64, 3, 7, 7) and c1.stride == (2, 2) and c1.padding == (3, 3) and c1.bias is not None
                and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1 and mp.dilation == 1 and not mp.ceil_mode)

    def forward(self, x):
        if self._stem_kernel_ok(x):
            # the whole stem -- cast, 7x7 convolution, bias, ReLU, max-pool -- in two launches of our own (csrc/stem.hip)
            from .ops import stem_conv_pool
            x = stem_conv_pool(x, self.conv1.weight, self.conv1.bias)
        elif getattr(self, "fused_epilogue", False) and x.is_cuda and x.dtype in (torch.float16, torch.float32):
            from .ops import bias_act_, bias_relu_maxpool
            _vendor_fallback("ResNet stem", routes.R.stem_kernel, f"7x7 convolution on {x.dtype} input of shape {tuple(x.shape)}")
            if x.dtype == torch.float32:
                x = x.half().contiguous(memory_format=torch.channels_last)
            x = F.conv2d(x, self.conv1.weight, None, self.conv1.stride, self.conv1.padding)
            mp = self.maxpool
            if (routes.R.stem_epilogue_kernel and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1 and mp.dilation == 1
                    and not mp.ceil_mode and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)):
                x = bias_relu_maxpool(x, self.conv1.bias)   # bias + ReLU + max-pool in one pass (bit-equal to the two)
            else:
                x = self.maxpool(bias_act_(x, self.conv1.bias, None, relu=True))
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class ConvModule(nn.Module):
    """mmcv ConvModule without norm/activation: parameters live under `.conv`."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)

    def forward(self, x):
        return self.conv(x)


@NECKS.register_module()
class FPN(BaseModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.start_level = start_level
        self.backbone_end_level = self.num_ins if end_level in (-1, self.num_ins - 1) else end_level + 1
        if num_outs != self.backbone_end_level - start_level:
            raise NotImplementedError("extra FPN levels are not used by the SimPB configs (num_outs == num_ins)")
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1))

    def forward(self, inputs):
        x0 = inputs[self.start_level]
        if (routes.R.conv1x1_kernel and x0.is_cuda and x0.dtype == torch.float16 and self.upsample_cfg.get("mode") == "nearest"
                and all(t.is_contiguous(memory_format=torch.channels_last) for t in inputs)
                and all(m.conv.in_channels % 64 == 0 and m.conv.out_channels % 8 == 0 for m in self.lateral_convs)):
            # lateral 1x1 convolutions as csrc/conv1x1.hip launches, the top-down sum riding along as the residual:
            # lateral[i-1] = conv(c[i-1]) + up(lateral[i])  (mmdet FPN.forward), coarsest level first
            from .ops import conv1x1_nhwc
            n = len(self.lateral_convs)
            laterals = [None] * n
            for i in range(n - 1, -1, -1):
                conv = self.lateral_convs[i].conv
                xin = inputs[i + self.start_level]
                up, up2x = None, False
                if i < n - 1:
                    hh, ww = xin.shape[2:]
                    if (hh % 2 == 0 and ww % 2 == 0 and tuple(laterals[i + 1].shape[2:]) == (hh // 2, ww // 2)
                            and "scale_factor" not in self.upsample_cfg):
                        up, up2x = laterals[i + 1], True  # nearest 2x read inside the launch: no upsampled map
                    else:
                        up = F.interpolate(laterals[i + 1], size=xin.shape[2:], **self.upsample_cfg)
                laterals[i] = conv1x1_nhwc(xin, conv.weight, conv.bias, up, relu=False, residual_upsample2x=up2x)
            tokens_for = getattr(self, "tokens_for", None)   # (bs, num_cams) set by SimPB.extract_feat for this call
            self.wrote_tokens = None
            if (routes.R.conv3x3_kernel and tokens_for is not None
                    and all(m.conv.in_channels % 64 == 0 and m.conv.out_channels % 8 == 0 and m.conv.padding == (1, 1)
                            and m.conv.stride == (1, 1) and m.conv.bias is not None for m in self.fpn_convs)):
                # the output convolutions write the decoder's fp32 token buffer themselves (csrc/conv3x3.hip): no f16 maps,
                # no format pass (feature_maps_format, ops/__init__.py:63-92)
                from .ops import conv3x3_nhwc, token_tables
                bs, num_cams = tokens_for
                shapes = tuple(tuple(t.shape[-2:]) for t in laterals)
                per_cam = sum(h * w for h, w in shapes)
                cout = self.fpn_convs[0].conv.out_channels
                col = torch.empty(bs, num_cams * per_cam, cout, device=x0.device, dtype=torch.float32)
                # the same rows without the widening: value_proj (group_attn.py:176) reads these (two-pass split product)
                col16 = torch.empty(bs, num_cams * per_cam, cout, device=x0.device, dtype=torch.float16)
                starts = [sum(h * w for h, w in shapes[:i]) for i in range(n)]
                if (routes.R.fpn_grouped_out and 1 < n <= 4 and len({m.conv.in_channels for m in self.fpn_convs}) == 1
                        and len({m.conv.out_channels for m in self.fpn_convs}) == 1):
                    # one launch for the four levels: the small levels' tiles fill the last round of the large one
                    from .ops import conv3x3_group_tokens
                    conv3x3_group_tokens(laterals, [m.conv.weight for m in self.fpn_convs], [m.conv.bias for m in self.fpn_convs],
                                         col, per_cam, starts, col16)
                else:
                    for i in range(n):
                        conv = self.fpn_convs[i].conv
                        conv3x3_nhwc(laterals[i], conv.weight, conv.bias, relu=False, tokens=(col, per_cam, starts[i], col16))
                col.simpb_f16 = col16
                self.deferred_output_bias = False
                self.wrote_tokens = [col, *token_tables(shapes, num_cams, col.device)]
                return ()
            _vendor_fallback("FPN output convolutions", routes.R.conv3x3_kernel and tokens_for is not None,
                             "3x3, " + ", ".join(f"{m.conv.in_channels}->{m.conv.out_channels}" for m in self.fpn_convs))
            self.deferred_output_bias = bool(getattr(self, "defer_output_bias", False))
            if self.deferred_output_bias:
                # the caller (SimPB.extract_feat) adds the biases while it writes the tokens (ops.format_tokens)
                return tuple(F.conv2d(laterals[i], self.fpn_convs[i].conv.weight, None, padding=1) for i in range(n))
            return tuple(self.fpn_convs[i](laterals[i]) for i in range(n))
        self.deferred_output_bias = False
        self.wrote_tokens = None
        _vendor_fallback("FPN lateral convolutions", routes.R.conv1x1_kernel and x0.is_cuda and x0.dtype == torch.float16,
                         "1x1, " + ", ".join(f"{m.conv.in_channels}->{m.conv.out_channels}" for m in self.lateral_convs))
        laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                              **self.upsample_cfg)
        return tuple(self.fpn_convs[i](laterals[i]) for i in range(len(laterals)))


@DETECTORS.register_module()
class SimPB(BaseModule):
    """models/simpb.py:25-129, inference path (simple_test)."""

    def __init__(self, img_backbone, head, img_neck=None, init_cfg=None, train_cfg=None, test_cfg=None,
                 pretrained=None, use_grid_mask=True, use_deformable_func=False, depth_branch=None):
        super().__init__(init_cfg)
        self.img_backbone = build_from_cfg(img_backbone, BACKBONES)
        if img_neck is not None:
            self.img_neck = build_from_cfg(img_neck, NECKS)
        else:
            self.img_neck = None
        self.head = build_from_cfg(head, HEADS)
        self.use_grid_mask = use_grid_mask  # GridMask is a training augmentation (grid_mask.py:93 returns x in eval)
        if not use_deformable_func:
            raise NotImplementedError("use_deformable_func=True is what the SimPB configs ship (config :55)")
        self.use_deformable_func = use_deformable_func
        self.head.use_deformable_func = use_deformable_func
        self.depth_branch = build_from_cfg(depth_branch, PLUGIN_LAYERS) if depth_branch is not None else None
        self.fp16_enabled = False

    def fuse_conv_bn(self):
        """tools/fuse_conv_bn.py:10-48 of the reference (its --fuse-conv-bn benchmark option): fold
        every eval-mode BatchNorm of the backbone into the convolution in front of it. Done in
        fp32, before any cast; the BatchNorm modules become Identity."""
        def fold(conv, bn):
            w = conv.weight
            b = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
            factor = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight = nn.Parameter(w * factor.reshape(-1, 1, 1, 1))
            conv.bias = nn.Parameter((b - bn.running_mean) * factor + bn.bias)

        bb = self.img_backbone
        with torch.no_grad():
            fold(bb.conv1, bb.bn1)
            bb.bn1 = nn.Identity()
            for name in bb.res_layers:
                for blk in getattr(bb, name):
                    for i in (1, 2, 3):
                        fold(getattr(blk, f"conv{i}"), getattr(blk, f"bn{i}"))
                        setattr(blk, f"bn{i}", nn.Identity())
                    if blk.downsample is not None:
                        fold(blk.downsample[0], blk.downsample[1])
                        blk.downsample[1] = nn.Identity()
        return self

    def half_backbone(self):
        """wrap_fp16_model (tools/test.py:239-241) + @auto_fp16(apply_to=('img',), out_fp32=True)
        (simpb.py:63): backbone and neck in fp16, everything after in fp32. channels_last so the
        convolutions produce NHWC, which is already the head's token layout."""
        self.img_backbone.half().to(memory_format=torch.channels_last)
        if self.img_neck is not None:
            self.img_neck.half().to(memory_format=torch.channels_last)
        self.fp16_enabled = True
        if isinstance(self.img_backbone.bn1, nn.Identity):  # BN folded: use the one-kernel conv epilogue
            self.img_backbone.fused_epilogue = True
            for name in self.img_backbone.res_layers:
                for blk in getattr(self.img_backbone, name):
                    blk.fused_epilogue = True
        return self

    def extract_feat(self, img, return_depth=False, metas=None):
        """simpb.py:64-91."""
        bs = img.shape[0]
        if img.dim() == 5:
            num_cams = img.shape[1]
            img = img.flatten(end_dim=1)
        else:
            num_cams = 1
        if self.fp16_enabled and not (img.dtype == torch.float32 and getattr(self.img_backbone, "_stem_kernel_ok", lambda t: False)(img)):
            img = img.half().contiguous(memory_format=torch.channels_last)   # (the stem kernel route casts inside its own launch)
        feature_maps = self.img_backbone(img)
        biases = None
        if self.img_neck is not None:
            # fp16 fused path: the four output convolutions of the FPN run without their bias and the bias is added
            # by the pass that writes the tokens (one launch for all levels instead of four bias launches + it)
            defer = (self.fp16_enabled and img.is_cuda and isinstance(self.img_neck, FPN)
                     and all(m.conv.bias is not None and m.conv.bias.dtype == torch.float16 and m.conv.padding == (1, 1)
                             and m.conv.out_channels % 8 == 0 for m in self.img_neck.fpn_convs))
            self.img_neck.defer_output_bias = defer
            self.img_neck.tokens_for = (bs, num_cams) if defer and feature_maps[0].shape[0] == bs * num_cams else None
            feature_maps = list(self.img_neck(feature_maps))
            if getattr(self.img_neck, "wrote_tokens", None) is not None:
                tokens, self.img_neck.wrote_tokens = self.img_neck.wrote_tokens, None
                return tokens
            if getattr(self.img_neck, "deferred_output_bias", False):  # (False when the neck took its unfused route)
                biases = [m.conv.bias for m in self.img_neck.fpn_convs]
        if feature_maps[0].is_cuda and feature_maps[0].shape[1] % 8 == 0:
            # channels_last maps are already token-major in memory: one conversion pass into col_feats
            return format_tokens(feature_maps, bs, num_cams, biases=biases)
        feature_maps = [f.float() if self.fp16_enabled else f for f in feature_maps]
        feature_maps = [torch.reshape(f, (bs, num_cams) + f.shape[1:]) for f in feature_maps]
        return feature_maps_format(feature_maps)

    def forward(self, img, **data):
        if self.training:
            raise NotImplementedError("training is out of scope of this path")
        return self.simple_test(img, **data)

    def simple_test(self, img, **data):
        feature_maps = self.extract_feat(img)
        model_outs = self.head(feature_maps, data)
        return self.head.post_process(model_outs, data)
