"""Soak of runner.SplitPipelinedRunner: N frames of synthetic features (the golden stream's generator) through the plain
and the split runner, every detection compared bit for bit; then the split runner again beside a second,
unrelated runner that keeps the chip busy. usage: python tools/split_soak.py [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import synth  # noqa: E402
from tests.helpers import load_golden, spec_of  # noqa: E402
from tests.test_gpu_runner import _golden_pipelined_runner  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
g = load_golden("head_r50.npz")
spec = spec_of(g)


_cache = {}


def stage(model, f):
    """Features of frame f % 8 (generated once on the CPU, kept on the GPU) into the model's staging buffer."""
    from simpb_amd.plugin import ops
    k = f % 8
    if k not in _cache:
        _cache[k] = ops.feature_maps_format([x.cuda() for x in synth.feature_maps_nchw(1, k, spec["image_wh"])])[0]
    if model.staged is None:
        model.staged = torch.empty_like(_cache[k])
    model.staged.copy_(_cache[k])


def run(split, busy=False):
    model, runner = _golden_pipelined_runner(spec, split)
    other = _golden_pipelined_runner(spec, False) if busy else None
    res = []
    for f in range(frames):
        stage(model, f)
        if other is not None:
            stage(other[0], f + 3)
        torch.cuda.synchronize()
        if f % 100 == 0:
            print("frame", f, flush=True)
        metas = synth.frame_metas(1, f, spec["image_wh"])
        runner.launch(runner.img, metas)
        if other is not None:
            other[1].launch(other[1].img, synth.frame_metas(1, f + 7, spec["image_wh"]))
        res.append(runner.collect())
        if other is not None:
            other[1].collect()
    res.append(runner.flush())
    print(("split runner" if split else "plain runner") + (" beside a second runner" if busy else ""), runner.stats, flush=True)
    return res[1:]


def same(a, b):
    bad = 0
    for f, (x, y) in enumerate(zip(a, b)):
        x, y = x[0]["img_bbox"], y[0]["img_bbox"]
        for k in x:
            u, v = (torch.as_tensor(np.asarray(t)) if not torch.is_tensor(t) else t.cpu() for t in (x[k], y[k]))
            if not torch.equal(u, v):
                bad += 1
                print("frame", f, k, "differs", flush=True)
                break
    return bad


ref = run(False)
print("split runner vs plain runner:", same(ref, run(True)), "frames differ of", frames)
print("split runner beside a second runner vs plain runner:", same(ref, run(True, busy=True)), "frames differ of", frames)
