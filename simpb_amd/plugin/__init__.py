"""Host-side mirror of projects/mmdet3d_plugin (models/__init__.py:1-44 is the export list).
Importing this package registers every class the shipped configs name."""
from .registry import (ATTENTION, BACKBONES, BBOX_CODERS, BBOX_SAMPLERS, DETECTORS, FEEDFORWARD_NETWORK, HEADS, LOSSES,
                       NECKS, NORM_LAYERS, PLUGIN_LAYERS, POSITIONAL_ENCODING, Config, build_from_cfg)
from .layers import MultiheadAttention
from .blocks import AsymmetricFFN, DeformableFeatureAggregation, DenseDepthNet
from .instance_bank import InstanceBank
from .detection2d import SparseBox2DEncoder, SparseBox2DRefinementModule
from .detection3d import (SparseBox3DDecoder, SparseBox3DEncoder, SparseBox3DKeyPointsGenerator,
                          SparseBox3DRefinementModule)
from .training_stubs import SparseBox3DTarget
from .allocation import DynamicQueryAllocation
from .aggregation import AdaptiveQueryAggregation
from .group_attn import QueryGroupMultiheadAttention, QueryGroupMultiScaleDeformableAttention
from .head import SimPBHead
from .detector import FPN, ResNet, SimPB
from .ops import deformable_aggregation_function, feature_maps_format

__all__ = [
    "SimPB", "SimPBHead", "DeformableFeatureAggregation", "DenseDepthNet", "AsymmetricFFN", "InstanceBank",
    "SparseBox3DDecoder", "SparseBox3DTarget", "SparseBox3DRefinementModule", "SparseBox3DKeyPointsGenerator",
    "SparseBox3DEncoder", "DynamicQueryAllocation", "AdaptiveQueryAggregation", "QueryGroupMultiheadAttention",
    "QueryGroupMultiScaleDeformableAttention",
]


def build_detector(cfg):
    return build_from_cfg(cfg, DETECTORS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)
