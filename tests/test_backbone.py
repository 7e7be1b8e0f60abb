"""CPU: conv-BN folding of the backbone leaves the FPN outputs unchanged (fp32)."""
import torch

from simpb_amd import configs, plugin, synth


def test_fuse_conv_bn_is_exact_in_fp32():
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    x = torch.from_numpy(synth.randn("bb.img", (2, 3, 64, 96)))
    with torch.no_grad():
        want = model.img_neck(model.img_backbone(x))
        model.fuse_conv_bn()
        got = model.img_neck(model.img_backbone(x))
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
