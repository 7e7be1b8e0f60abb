"""CPU: the boundary. The reference's shipped configs load UNCHANGED through simpb_amd's registry
(only where /root/reference exists, i.e. the build container), simpb_amd.configs restates their
`model` dict exactly, and the built head exposes the reference's state_dict keys and shapes."""
import os

import numpy as np
import pytest

from simpb_amd import configs, plugin, synth
from tests.helpers import load_golden

REF_CFG = "/root/reference/projects/configs"


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("name,pretrained", [
    ("simpb_nus_r50_img_704x256.py", "ckpts/resnet50-19c8e357.pth"),
    ("simpb_nus_r50_uimg_704x256.py", None),
])
def test_reference_config_loads_unchanged(name, pretrained):
    cfg = plugin.Config.fromfile(os.path.join(REF_CFG, name))
    mine = configs.simpb_plus()
    ref_model = _plain(cfg.model)
    my_model = _plain(mine["model"])
    if pretrained is None:  # the nuImages variant differs only in how the backbone is initialised (:79-99,460-463)
        ref_model["img_backbone"] = {k: v for k, v in ref_model["img_backbone"].items() if k not in ("pretrained", "init_cfg")}
        my_model["img_backbone"] = {k: v for k, v in my_model["img_backbone"].items() if k not in ("pretrained", "init_cfg")}
    assert ref_model == my_model
    cfg.merge_from_dict({"model.head.instance_bank.anchor": synth.anchors(900)})  # the k-means file is not offline
    model = plugin.build_detector(cfg.model)
    assert type(model).__name__ == "SimPB" and len(model.head.layers) == 50


def test_head_state_dict_matches_reference_keys():
    g = load_golden("head_r50.npz")
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    head = plugin.build_head(cfg["model"]["head"])
    want = dict(zip(g["state_keys"].tolist(), g["state_shapes"].tolist()))
    got = {k: ",".join(map(str, v.shape)) for k, v in head.state_dict().items()}
    assert got == want
    assert head.operation_order == g["operation_order"].tolist()
    counts = {op: head.operation_order.count(op) for op in set(head.operation_order)}
    assert counts == dict(allocation=3, qg_self_attn=3, norm=12, qg_cross_attn=3, ffn=6, refine2d=3, aggregation=3,
                          refine3d=6, temp_gnn=5, gnn=3, deformable=3)


def test_registry_surface():
    """every registry name the shipped configs use resolves."""
    for reg, names in [
        (plugin.DETECTORS, ["SimPB"]), (plugin.HEADS, ["SimPBHead"]),
        (plugin.ATTENTION, ["DeformableFeatureAggregation", "QueryGroupMultiheadAttention",
                            "QueryGroupMultiScaleDeformableAttention", "MultiheadAttention"]),
        (plugin.PLUGIN_LAYERS, ["DynamicQueryAllocation", "AdaptiveQueryAggregation", "InstanceBank",
                                "SparseBox3DRefinementModule", "SparseBox2DRefinementModule",
                                "SparseBox3DKeyPointsGenerator", "DenseDepthNet", "Denoise2D"]),
        (plugin.POSITIONAL_ENCODING, ["SparseBox3DEncoder", "SparseBox2DEncoder"]),
        (plugin.FEEDFORWARD_NETWORK, ["AsymmetricFFN"]), (plugin.NORM_LAYERS, ["LN"]),
        (plugin.BBOX_CODERS, ["SparseBox3DDecoder"]), (plugin.BBOX_SAMPLERS, ["SparseBox3DTargetWith2D", "SparseBox2DCoster"]),
        (plugin.LOSSES, ["FocalLoss", "L1Loss", "GIoULoss", "SparseBox3DLoss", "CrossEntropyLoss", "GaussianFocalLoss"]),
        (plugin.BACKBONES, ["ResNet"]), (plugin.NECKS, ["FPN"]),
    ]:
        for n in names:
            assert n in reg, (reg.name, n)


def test_product_path_has_no_cpu_fallback():
    """The head's operators refuse CPU tensors instead of silently computing somewhere else."""
    import torch
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    head = plugin.build_head(cfg["model"]["head"]).eval()
    fm = plugin.feature_maps_format([x[:, :, :8] for x in synth.feature_maps_nchw(1, 0, image_wh=(88, 32))])
    with pytest.raises(RuntimeError), torch.no_grad():
        head([fm[0].repeat(1, 1, 32), fm[1], fm[2]], synth.frame_metas(1, 0, image_wh=(88, 32)))
