"""Registry entries the shipped configs name but which only act in training (SURVEY.md §2a rows
17-19: Hungarian targets, denoising, losses). They are registered so that `projects/configs/*.py`
build unchanged; each keeps its constructor kwargs and fails loudly if its training entry point
is called. `sampler.dn_metas` is read in eval (simpb_head.py:333), hence the attribute."""
from .registry import BBOX_SAMPLERS, LOSSES, PLUGIN_LAYERS


class _Inert:
    def __init__(self, *args, **kwargs):
        self.cfg = dict(kwargs)

    def _no(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is training-only; this build covers the inference hot path")

    __call__ = forward = loss = sample = get_dn_anchors = update_dn = cache_dn = _no


class _Target(_Inert):
    def __init__(self, *args, num_dn_groups=0, num_temp_dn_groups=0, **kwargs):
        super().__init__(**kwargs)
        self.num_dn_groups = num_dn_groups
        self.num_temp_dn_groups = num_temp_dn_groups
        self.dn_metas = None


@BBOX_SAMPLERS.register_module()
class SparseBox3DTarget(_Target):
    """models/detection3d/target.py:57-431."""


@BBOX_SAMPLERS.register_module()
class SparseBox3DTargetWith2D(_Target):
    """models/detection3d/target.py:432-965."""


@BBOX_SAMPLERS.register_module()
class SparseBox2DTarget(_Target):
    """models/detection2d/target.py."""


@BBOX_SAMPLERS.register_module()
class SparseBox2DCoster(_Inert):
    """models/detection2d/coster.py."""


@PLUGIN_LAYERS.register_module()
class Denoise2D(_Inert):
    """models/detection2d/denoise.py:9-228 (every use sits behind `dn_metas is not None`)."""

    def __init__(self, num_cams=6, num_dn_groups=0, with_attn_mask=False):
        super().__init__(num_cams=num_cams, num_dn_groups=num_dn_groups, with_attn_mask=with_attn_mask)
        self.num_cams, self.num_dn_groups, self.with_attn_mask = num_cams, num_dn_groups, with_attn_mask


for _name in ("FocalLoss", "L1Loss", "GIoULoss", "CrossEntropyLoss", "GaussianFocalLoss", "SmoothL1Loss"):
    LOSSES.register_module(_name, module=type(_name, (_Inert,), {"__doc__": "mmdet loss (training only)"}))


@LOSSES.register_module()
class SparseBox3DLoss(_Inert):
    """models/detection3d/losses.py:11-69."""
