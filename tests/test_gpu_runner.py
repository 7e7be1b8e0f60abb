"""GPU: the static-shape / hipGraph frame runner produces the reference's detections. Same golden
streams as test_gpu_head.py, but through simpb_amd.runner.FrameRunner (fixed-capacity 2D query set,
device-side group table, persistent bank buffers, captured warm frame)."""
import json
import os

import numpy as np
import pytest
import torch

from simpb_amd import synth
from tests.helpers import build_product_head, compare_result, load_golden, spec_of

pytestmark = pytest.mark.gpu


class _FeatureModel(torch.nn.Module):
    """Stand-in for the detector around the head: `extract_feat` returns the synthetic feature maps
    of the golden stream (the golden vectors were captured from the head alone)."""

    def __init__(self, head, spec):
        super().__init__()
        self.head = head
        self.spec = spec
        self.frame = 0
        self.maps = None

    def load(self, f):
        from simpb_amd.plugin import ops
        fm = ops.feature_maps_format([x.cuda() for x in synth.feature_maps_nchw(self.spec["bs"], f, self.spec["image_wh"])])
        if self.maps is None:
            self.maps = fm
        else:
            self.maps[0].copy_(fm[0])  # fixed address: the captured frame reads this buffer

    def extract_feat(self, img):
        return self.maps


@pytest.mark.parametrize("name,capacity,use_graph", [
    ("head_small.npz", 96, True), ("head_r50.npz", 1536, True), ("head_r50.npz", 1280, False)])
def test_runner_stream_vs_golden(name, capacity, use_graph):
    from simpb_amd.runner import FrameRunner
    g = load_golden(name)
    spec = spec_of(g)
    head = build_product_head(spec)
    model = _FeatureModel(head, spec)
    w, h = spec["image_wh"]
    runner = FrameRunner(model, spec["bs"], (8, 8), capacity=capacity, device=torch.device("cuda"), use_graph=use_graph)
    runner.wh = torch.tensor([float(w), float(h)], device="cuda").view(1, 1, 2).repeat(spec["bs"], 6, 1)
    runner.wh_host = (w, h)
    frames = spec["frames"] + (2 if use_graph else 0)
    for f in range(spec["frames"]):
        model.load(f)
        metas = synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"])
        res = runner.step(runner.img, metas)
        for b, r in enumerate(res):
            compare_result(r["img_bbox"], g, f"f{f}.res{b}.")
    if use_graph:
        assert runner.stats["replay"] >= 1, runner.stats


@pytest.mark.parametrize("name,capacity,use_graph", [("head_small.npz", 16, False), ("head_r50.npz", 512, True)])
def test_runner_overflow_reruns_the_frame(name, capacity, use_graph):
    """A 2D query set larger than the static capacity: the runner grows the capacity and re-runs that frame on the
    bank state the frame found (the frame-end commit holds back while an overflow flag is set), re-captures its graph,
    and the stream's detections, track ids included, are the golden ones."""
    from simpb_amd.runner import FrameRunner
    g = load_golden(name)
    spec = spec_of(g)
    model = _FeatureModel(build_product_head(spec), spec)
    runner = FrameRunner(model, spec["bs"], (8, 8), capacity=capacity, device=torch.device("cuda"), use_graph=use_graph)
    w, h = spec["image_wh"]
    runner.wh = torch.tensor([float(w), float(h)], device="cuda").view(1, 1, 2).repeat(spec["bs"], 6, 1)
    runner.wh_host = (w, h)
    for f in range(spec["frames"]):
        model.load(f)
        res = runner.step(runner.img, synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"]))
        for b, r in enumerate(res):
            compare_result(r["img_bbox"], g, f"f{f}.res{b}.")
    assert runner.stats["overflow"] >= 1 and runner.capacity > capacity, (runner.stats, runner.capacity)


def test_runner_overflow_after_capture_reruns_and_recaptures():
    """Overflow in the middle of a warm stream: the golden R50 stream runs two frames at a capacity that fits, then the
    slot array is shrunk between frames, so frame 2 overflows with a warm bank, is re-run at a grown capacity on the
    state it found, and the rest of the stream (re-captured graph) still matches the golden detections and ids."""
    from simpb_amd.runner import FrameRunner
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    model = _FeatureModel(build_product_head(spec), spec)
    runner = FrameRunner(model, 1, (8, 8), capacity=1536, device=torch.device("cuda"), use_graph=True)
    w, h = spec["image_wh"]
    runner.wh = torch.tensor([float(w), float(h)], device="cuda").view(1, 1, 2).repeat(1, 6, 1)
    runner.wh_host = (w, h)
    for f in range(spec["frames"]):
        if f == 2:  # between frames: shrink the slot array; the frame that follows overflows, grows, re-runs
            runner.capacity = runner.head.static_capacity = 640
            runner._drop_graphs()
        model.load(f)
        res = runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"], jump=spec["jump"]))
        compare_result(res[0]["img_bbox"], g, f"f{f}.res0.")
    assert runner.stats["overflow"] >= 1 and runner.stats["replay"] >= 1, runner.stats


class _StagedModel(torch.nn.Module):
    """Detector stand-in for the pipelined runner: the "image" is ignored, features are the golden stream's
    synthetic maps, copied from one persistent staging buffer (a captured backbone graph bakes in the source
    address) into a per-slot buffer so that the captured graphs stay valid."""

    def __init__(self, head, spec):
        super().__init__()
        self.head = head
        self.spec = spec
        self.bufs = {}
        self.staged = None

    def stage(self, f):
        from simpb_amd.plugin import ops
        fresh = ops.feature_maps_format([x.cuda() for x in synth.feature_maps_nchw(self.spec["bs"], f, self.spec["image_wh"])])[0]
        if self.staged is None:
            self.staged = torch.empty_like(fresh)
        self.staged.copy_(fresh)

    def extract_feat(self, img):
        from simpb_amd.plugin import ops
        key = img.data_ptr()
        if key not in self.bufs:
            self.bufs[key] = ops.feature_maps_format(
                [torch.zeros_like(x).cuda() for x in synth.feature_maps_nchw(self.spec["bs"], 0, self.spec["image_wh"])])
        self.bufs[key][0].copy_(self.staged, non_blocking=True)
        return self.bufs[key]


def _golden_pipelined_runner(spec, split=False, capacity=1536):
    from simpb_amd.runner import PipelinedRunner, SplitPipelinedRunner
    model = _StagedModel(build_product_head(spec), spec)
    w, h = spec["image_wh"]
    bs = spec["bs"]
    runner = (SplitPipelinedRunner if split else PipelinedRunner)(model, bs, (8, 8), capacity=capacity,
                                                                  device=torch.device("cuda"), use_graph=True)
    runner.wh = torch.tensor([float(w), float(h)], device="cuda").view(1, 1, 2).repeat(bs, 6, 1)
    runner.wh_host = (w, h)
    return model, runner


@pytest.mark.parametrize("split", [False, True])
def test_pipelined_runner_vs_golden(split):
    """Backbone(t+1) overlapped with decoder(t): same detections, one step later. split: the single-frame decoder layer of
    frame t+1 additionally runs beside the temporal part of frame t (runner.SplitPipelinedRunner, SimPBHead.forward_split)."""
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    model, runner = _golden_pipelined_runner(spec, split)
    outs = []
    for f in range(spec["frames"]):
        model.stage(f)
        torch.cuda.synchronize()
        outs.append(runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"])))
    outs.append(runner.flush())
    assert outs[0] is None
    for f in range(spec["frames"]):
        compare_result(outs[f + 1][0]["img_bbox"], g, f"f{f}.res0.")


@pytest.mark.parametrize("split", [False, True])
def test_pipelined_runner_batched_stream_with_a_time_gap(split):
    """The small golden stream (bs = 2, one stream jumps in time: the bank's max_time_interval mask, instance_bank.py:87,
    and the default time step in the refinement heads, :108-113) through both pipelined runners. In the split runner the
    time step of the single-frame layer is derived on the host from the time stamps (runner._stage_slot)."""
    g = load_golden("head_small.npz")
    spec = spec_of(g)
    model, runner = _golden_pipelined_runner(spec, split, capacity=96)
    outs = []
    for f in range(spec["frames"]):
        model.stage(f)
        torch.cuda.synchronize()
        outs.append(runner.step(runner.img, synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"])))
    outs.append(runner.flush())
    for f in range(spec["frames"]):
        for b in range(spec["bs"]):
            compare_result(outs[f + 1][b]["img_bbox"], g, f"f{f}.res{b}.")


@pytest.mark.parametrize("split", [False, True])
def test_pipelined_runner_overflow_reruns_the_decoder(split):
    """Same stream through the pipelined runner with a slot array that is too small: the decoder of the overflowed
    frame is re-run (beside the next frame's backbone) and the golden detections come out, one step later."""
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    model, runner = _golden_pipelined_runner(spec, split)
    runner.capacity = runner.head.static_capacity = 768
    outs = []
    for f in range(spec["frames"]):
        model.stage(f)
        torch.cuda.synchronize()
        outs.append(runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"])))
    outs.append(runner.flush())
    for f in range(spec["frames"]):
        compare_result(outs[f + 1][0]["img_bbox"], g, f"f{f}.res0.")
    assert runner.stats["overflow"] >= 1 and runner.capacity > 768, (runner.stats, runner.capacity)


@pytest.mark.parametrize("split", [False, True])
def test_pipelined_runner_overflow_with_a_decoder_already_enqueued_behind(split):
    """Overflow in the middle of a warm stream of the pipelined runner: the slot array is shrunk before frame 1 is fed,
    so decoder(1) overflows with a warm bank WHILE decoder(2) has already been enqueued behind it (the host had not seen
    the flags yet). The bank commit of both holds back on the device (overflow_chain), both are re-run in order at a
    grown capacity, and the stream's detections and track ids are the golden ones."""
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    model, runner = _golden_pipelined_runner(spec, split)
    outs = []
    for f in range(spec["frames"]):
        if f == 1:
            runner.capacity = runner.head.static_capacity = 640
            runner._drop_graphs()
        model.stage(f)
        torch.cuda.synchronize()
        outs.append(runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"], jump=spec["jump"])))
    outs.append(runner.flush())
    assert outs[0] is None and not runner.queue
    for f in range(spec["frames"]):
        compare_result(outs[f + 1][0]["img_bbox"], g, f"f{f}.res0.")
    assert runner.stats["overflow"] >= 1 and runner.capacity > 640, (runner.stats, runner.capacity)
    assert int(runner.flags.abs().sum()) == 0
    if split:
        assert int(runner.sticky.item()) == 0 and int(runner.hb.abs().sum()) == 0


@pytest.mark.parametrize("split", [False, True])
def test_overflow_with_a_time_gap_frame_enqueued_behind_it(split):
    """The chained hold and the track ids: frame 1 of the small golden stream overflows its (shrunk) slot array while the
    decoder of frame 2 is already enqueued behind it -- and frame 2 is the one where stream 1 jumps in time, i.e. whose
    InstanceBank.update resets that stream's track ids (instance_bank.py:147-149). That speculative frame must leave the ids
    alone (the reset is gated by the hold, csrc/bank.hip): both frames are re-run in order on the state frame 0 left, and
    every frame's detections and ids are the golden ones."""
    g = load_golden("head_small.npz")
    spec = spec_of(g)
    assert spec["jump"] is not None and spec["jump"][1] == 2
    model, runner = _golden_pipelined_runner(spec, split, capacity=96)
    outs = []
    for f in range(spec["frames"]):
        if f == 1:
            runner.capacity = runner.head.static_capacity = 16   # frame 1 (and the speculative frame 2) overflow
            runner._drop_graphs()
        model.stage(f)
        torch.cuda.synchronize()
        outs.append(runner.step(runner.img, synth.frame_metas(spec["bs"], f, spec["image_wh"], jump=spec["jump"])))
    outs.append(runner.flush())
    assert outs[0] is None and not runner.queue
    assert runner.stats["overflow"] >= 1 and runner.capacity > 16, (runner.stats, runner.capacity)
    for f in range(spec["frames"]):
        for b in range(spec["bs"]):
            compare_result(outs[f + 1][b]["img_bbox"], g, f"f{f}.res{b}.")
    assert int(runner.flags.abs().sum()) == 0
    if split:
        assert int(runner.sticky.item()) == 0 and int(runner.hb.abs().sum()) == 0


def test_split_runner_equals_two_stream_runner_bit_for_bit():
    """40 frames of synthetic features through PipelinedRunner and SplitPipelinedRunner (graphs, part A of
    frame t beside part B of frame t-1): the same launches on the same numbers, so every detection, score and track id must
    be IDENTICAL, frame by frame -- a stale staging buffer, a flag row touched by the wrong frame or a part B that started
    before its part A would show here."""
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    frames = 40
    outs = {}
    for split in (False, True):
        model, runner = _golden_pipelined_runner(spec, split)
        res = []
        for f in range(frames):
            model.stage(f)
            torch.cuda.synchronize()
            res.append(runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"])))
        res.append(runner.flush())
        assert runner.stats["replay"] >= frames - 10 and runner.stats["overflow"] == 0, runner.stats
        outs[split] = res[1:]
    for f, (a, b) in enumerate(zip(outs[False], outs[True])):
        a, b = a[0]["img_bbox"], b[0]["img_bbox"]
        assert set(a) == set(b)
        for k in a:
            x, y = (torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v for v in (a[k], b[k]))
            assert torch.equal(x.cpu(), y.cpu()), (f, k)


@pytest.mark.parametrize("split", [False, True])
def test_config3_eight_streams_per_gpu_vs_golden(split):
    """BASELINE config #3 shape on one GPU: 8 independent camera streams at R50 704x256, each its own pipelined runner
    (bs = 1, the reference's own test setting), launched back to back and collected together like bench.py --streams 8.
    Every stream is fed the golden feature stream (stream i starts i % 3 steps late, so neighbours are at different
    frames) and must return the golden detections; the device records of the eight streams go through
    dist.DetectionGather (the exchange bench.py runs over RCCL) and must equal the host results bit for bit, int64
    track ids included."""
    from simpb_amd.dist import DetectionGather, unpack_detections
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    n, frames = 8, spec["frames"]
    pairs = [_golden_pipelined_runner(spec, split) for _ in range(n)]
    lag = [i % 3 for i in range(n)]
    gather = DetectionGather(n, 300, torch.device("cuda"))
    outs = [[] for _ in range(n)]
    fed = [0] * n
    checked = 0
    for step in range(frames + max(lag)):
        live = [i for i in range(n) if 0 <= step - lag[i] < frames]
        for i in live:
            model, runner = pairs[i]
            model.stage(step - lag[i])
            torch.cuda.synchronize()  # the staging buffer is test scaffolding shared with the runner's streams
            runner.launch(runner.img, synth.frame_metas(1, step - lag[i], spec["image_wh"]))
        for i in live:
            outs[i].append(pairs[i][1].collect())
        if len(live) == n and all(o[-1] is not None for o in outs):
            gather.submit([pairs[i][1].last_rec3d for i in range(n)], [pairs[i][1].s_rec for i in range(n)])
            for i in range(n):
                pairs[i][1].rec_consumed = gather.done
            rec = unpack_detections(gather.result()[0])
            for i in range(n):
                want = outs[i][-1][0]["img_bbox"]
                assert torch.equal(rec["boxes_3d"][i], want["boxes_3d"]) and torch.equal(rec["scores_3d"][i], want["scores_3d"])
                assert torch.equal(rec["instance_ids"][i], want["instance_ids"]) and torch.equal(rec["labels_3d"][i], want["labels_3d"])
            checked += 1
    assert checked >= 1
    for i in range(n):
        out = outs[i][1:] + [pairs[i][1].flush()]
        for f in range(frames):
            compare_result(out[f][0]["img_bbox"], g, f"f{f}.res0.")


@pytest.mark.parametrize("compact", [False, True], ids=["slot_array", "compacted_rows"])
@pytest.mark.parametrize("split", [False, True])
def test_gathered_record_survives_two_more_steps_before_it_is_read(split, compact):
    """The exchange's asynchronous window: the device records (3D and 2D) of the frame step(t) returned are SUBMITTED to
    dist.DetectionGather without waiting for the exchange, the runner goes on for two more steps (replayed graphs: part
    A of frame t+2 shares part B's memory pool and is enqueued on the other stream), and only then the gathered record is
    read: it must still be frame t's, bit for bit. Each runner makes every graph that may write the record's memory wait
    for `rec_consumed`; here the side stream is held up behind a long kernel so that the copy really is still pending
    when those graphs are enqueued."""
    from simpb_amd.dist import DetectionGather, unpack_detections, unpack_detections2d
    g = load_golden("head_r50.npz")
    spec = spec_of(g)
    model, runner = _golden_pipelined_runner(spec, split)
    # compact: the exchange bench.py uses -- only the rows of the kept 3D boxes travel (csrc/decode.hip record2d_compact_kernel),
    # 300 x 6 rows whatever the runner's slot capacity is; otherwise the slot array itself (900 x 6 rows)
    gather = DetectionGather(1, 300, torch.device("cuda"), rows2d=300 * 6 if compact else 900 * 6, compact2d=compact)
    spin = torch.empty(64 * 1024 * 1024, device="cuda")
    frames, pending, checked = 30, [], 0
    for f in range(frames):
        model.stage(f)
        torch.cuda.synchronize()
        res = runner.step(runner.img, synth.frame_metas(1, f, spec["image_wh"]))
        if pending and f - pending[0][0] >= 2:     # two steps later: read what was submitted then
            _, want = pending.pop(0)
            rec = unpack_detections(gather.result()[0])
            assert torch.equal(rec["boxes_3d"][0], want["boxes_3d"]) and torch.equal(rec["scores_3d"][0], want["scores_3d"])
            assert torch.equal(rec["instance_ids"][0], want["instance_ids"])
            got2d = unpack_detections2d(gather.result2d()[0])[0]
            assert torch.equal(got2d["boxes_2d"], want["boxes_2d"]) and torch.equal(got2d["scores_2d"], want["scores_2d"])
            assert torch.equal(got2d["labels_2d"], want["labels_2d"]) and torch.equal(got2d["camidx_2d"], want["camidx_2d"])
            if compact:   # kept rows in front, nothing but pad rows behind, and the host decode of the compacted record = the frame's
                raw = gather.result2d()[0, 0].cpu()
                n = len(want["boxes_2d"])
                assert bool((raw[:n, 6] >= 0).all()) and bool((raw[n:, 6:8] == -1).all()) and float(raw[n:, :6].abs().max()) == 0.0
            checked += 1
        if res is not None and not pending and f >= 8:   # replayed frames only, one exchange in flight at a time
            with torch.cuda.stream(gather.side):   # keep the side stream busy: the copy below stays pending for a while
                for _ in range(20):
                    spin.add_(1.0)
            gather.submit([runner.last_rec3d], [runner.s_rec], records2d=[runner.last_rec2d])
            runner.rec_consumed = gather.done
            pending.append((f, res[0]["img_bbox"]))
    assert checked >= 5 and runner.stats["replay"] >= frames - 10, (checked, runner.stats)


class _ReplayModel(torch.nn.Module):
    """Detector stand-in that serves recorded feature maps from fixed-address buffers."""

    def __init__(self, head):
        super().__init__()
        self.head = head
        self.maps = None

    def load(self, fm):
        if self.maps is None:
            self.maps = [t.clone() for t in fm]
        else:
            for dst, src in zip(self.maps, fm):
                dst.copy_(src)

    def extract_feat(self, img):
        return self.maps


def test_pipelined_equals_plain_runner_with_real_backbone():
    """Whole detector (ResNet50+FPN with folded BN + decoder), 12 frames, through the graph runner and the
    pipelined runner (two streams, four graphs, backbone(t+1) beside decoder(t)), the latter also with the frames
    arriving in pinned host memory.

    The vendor's convolutions are not bit-reproducible from one run to the next (tools/pipe_determinism.py:
    the feature maps of two EAGER runs already differ in every frame), and a 1e-7 difference that flips one
    inside/outside test of the query allocation moves scores by 1e-2 for the rest of the stream, so two runs
    of the detector cannot be compared to 1e-3. Instead each runner's features are recorded as its decoder
    saw them and served again to the plain eager runner: a stale slot, a decoder that started before its
    backbone finished or a backbone that overwrote maps still in use would all show as a difference."""
    from simpb_amd import configs, plugin
    from simpb_amd.runner import FrameRunner, PipelinedRunner, SplitPipelinedRunner
    wh = (352, 128)

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn()

    frames = 12
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]

    def snapshot(fm):
        return [t.clone() for t in list(fm)[:3]]

    for name in ("graph", "pipe", "pipe_h2d", "pipe_split"):   # pipe_split: + the single-frame decoder layer beside the previous frame's temporal part
        model = make()
        seen = []
        if name.startswith("pipe"):
            cls = SplitPipelinedRunner if name == "pipe_split" else PipelinedRunner
            r = cls(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True)
            # pipe_h2d: frames handed over in pinned host memory, copied inside the step on the backbone stream
            # beside the previous frame's decoder (bench.py --h2d)
            src = imgs if name != "pipe_h2d" else [x.cpu().pin_memory() for x in imgs]
            out = []
            for f in range(frames):
                out.append(r.step(src[f], metas[f]))
                r.s_bb.synchronize()  # (the test reads the runner's buffers from another stream; step() does not wait for backbone(f))
                assert torch.equal(r.imgs[f % 2].cpu(), imgs[f].cpu()), (name, f)  # the frame the backbone just read
                if f >= 1:  # the maps decoder(f-1) just read; backbone(f) wrote the other slot meanwhile
                    seen.append(snapshot(r.fm[(f - 1) % 2]))
            seen.append(snapshot(r.fm[(frames - 1) % 2]))
            out = out[1:] + [r.flush()]
            assert r.stats["replay"] >= 4, r.stats
        else:
            r = FrameRunner(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True)
            inner, last = model.extract_feat, {}

            def spy(img):
                last["fm"] = inner(img)
                return last["fm"]

            model.extract_feat = spy
            out = []
            for f in range(frames):
                out.append(r.step(imgs[f], metas[f]))
                seen.append(snapshot(last["fm"]))
            assert r.stats["replay"] >= 4, r.stats
        _check_against_replay(make().head, seen, out, metas, wh, name)


def _check_against_replay(head, seen, out, metas, wh, name):
    """Serve the recorded feature maps `seen[f]` to the plain eager runner and compare with `out[f]`."""
    from simpb_amd.runner import FrameRunner
    replay = _ReplayModel(head)
    plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=False)
    for f in range(len(seen)):
        replay.load(seen[f])
        a = plain.step(plain.img, metas[f])[0]["img_bbox"]
        b = out[f][0]["img_bbox"]
        assert a["boxes_3d"].shape == b["boxes_3d"].shape
        assert float((a["scores_3d"] - b["scores_3d"]).abs().max()) <= 1e-3, (name, f)
        assert rows_match_t(a["boxes_3d"], b["boxes_3d"], 1e-3), (name, f)


@pytest.mark.parametrize("bs", [3])
def test_pipelined_batch_of_independent_streams_equals_the_plain_eager_batch(bs):
    """BASELINE config #3 in its throughput form ('keep per-sample counts in the native path'): bs camera
    streams with different headings and time origins through ONE runner launch per frame (independent_streams=True: one
    flat 2D slot array over bs x 6 camera groups, csrc/alloc.hip alloc_scatter_ragged_kernel), images through the real
    backbone, eight frames with the temporal bank, one stream jumping in time. The pipelined runner (two streams, replayed
    graphs, backbone(t+1) beside decoder(t)) must return EXACTLY what the plain eager runner of the same batch returns for
    the features the pipelined decoder saw (recorded and served again, as in
    test_pipelined_equals_plain_runner_with_real_backbone): same kernels, same layout, so bit for bit -- a stale slot, a
    decoder ahead of its backbone or a graph replayed on the wrong tables would show. That the batch equals one runner PER
    STREAM is the next test's claim (it cannot be made bit for bit: the flat layout shifts the attention key tiles)."""
    from simpb_amd import configs, plugin
    from simpb_amd.runner import FrameRunner, PipelinedRunner
    wh = (352, 128)

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn().half_backbone()

    frames = 8
    imgs = [synth.images(bs, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(bs, f, wh, jump=(1, 5, 10.0)) for f in range(frames)]
    batch = PipelinedRunner(make(), bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True,
                            independent_streams=True)
    seen, got = [], []
    for f in range(frames):
        got.append(batch.step(imgs[f], metas[f]))
        batch.s_bb.synchronize()
        if f >= 1:
            seen.append([t.clone() for t in list(batch.fm[(f - 1) % 2])[:3]])
    seen.append([t.clone() for t in list(batch.fm[(frames - 1) % 2])[:3]])
    got = got[1:] + [batch.flush()]
    assert batch.stats["replay"] >= 2 and batch.stats["overflow"] == 0, batch.stats
    assert tuple(batch.last_rec2d.shape) == (bs, 1536, 8) and tuple(batch.last_rec3d.shape) == (bs, 300, 15)
    replay = _ReplayModel(make().head)
    plain = FrameRunner(replay, bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=False,
                        independent_streams=True)
    for f in range(frames):
        replay.load(seen[f])
        want = plain.step(plain.img, metas[f])
        for b in range(bs):
            x, y = got[f][b]["img_bbox"], want[b]["img_bbox"]
            assert set(x) == set(y)
            for k in ("boxes_3d", "scores_3d", "labels_3d", "cls_scores", "instance_ids", "boxes_2d", "scores_2d", "labels_2d",
                      "camidx_2d", "trans_matrix"):
                assert torch.equal(torch.as_tensor(np.asarray(x[k])), torch.as_tensor(np.asarray(y[k]))), (f, b, k)
            assert x["query_groups"] == y["query_groups"]


def test_batch_of_independent_streams_overflow_reruns_and_matches_a_roomy_run():
    """A stream that needs more 2D slots than the per-stream capacity sets the overflow flag (csrc/alloc.hip clips it like a
    batch of one); the runner re-runs the frame with a larger slot array on the untouched bank, as for a single stream. The
    flat layout packs live slots first, so the capacity changes no live slot's position: the detections must EQUAL those of
    a run that had room from the start, bit for bit (same images, same batch, deterministic kernels)."""
    from simpb_amd import configs, plugin
    from simpb_amd.runner import PipelinedRunner
    wh, bs, frames = (352, 128), 2, 6

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn().half_backbone()

    imgs = [synth.images(bs, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(bs, f, wh) for f in range(frames)]
    outs = {}
    for cap in (256, 1536):
        r = PipelinedRunner(make(), bs, (wh[1], wh[0]), capacity=cap, device=torch.device("cuda"), use_graph=True,
                            independent_streams=True)
        out = [r.step(imgs[f], metas[f]) for f in range(frames)]
        outs[cap] = out[1:] + [r.flush()]
        if cap == 256:
            assert r.stats["overflow"] >= 1 and r.capacity > 256, (r.stats, r.capacity)
        else:
            assert r.stats["overflow"] == 0
    for f in range(frames):
        for b in range(bs):
            a, c = outs[256][f][b]["img_bbox"], outs[1536][f][b]["img_bbox"]
            for k in ("boxes_3d", "scores_3d", "labels_3d", "boxes_2d", "scores_2d", "labels_2d", "camidx_2d", "instance_ids"):
                assert torch.equal(torch.as_tensor(np.asarray(a[k])), torch.as_tensor(np.asarray(c[k]))), (f, b, k)


def _matched_fraction(got, want, tol):
    """Fraction of the rows of `want` that have a row of `got` within tol (max norm over the columns)."""
    got, want = (torch.as_tensor(np.asarray(x, np.float64)) for x in (got, want))
    if want.shape[0] == 0 or got.shape[0] == 0:
        return 1.0 if want.shape[0] == got.shape[0] else 0.0
    return float((torch.cdist(want, got, p=float("inf")).min(dim=1).values <= tol).double().mean())


def test_batch_of_independent_streams_frame_by_frame_from_the_same_state():
    """A batch of independent streams must return, stream by stream, what a runner of batch one returns for that stream:
    bs = 4, real backbone, replayed graph, one stream jumping in time, checked frame by frame FROM THE SAME STATE: before
    frame f the plain batch-of-one runner of stream b is handed the bank state (cached features / anchors / confidences /
    track ids) the batched runner held for that stream, so a difference can only come from this frame's arithmetic.
    Why not the whole stream: round 4 traced round 3's red whole-stream comparison (stream 1 around its jump frame)
    operator by operator (tools/diag_ragged_trace.py, profiles/r04_ragged_trace_bs3.log): from the same state the flat layout
    and the batch of one agree to <= 3e-5 of each record's scale on every operator boundary and on the state they leave
    behind, jump frame included; what separates two whole streams is the random-weight decoder's sensitivity (one 2D
    self-attention layer turns a 4e-5 input difference into 2e-3 on every row -- between two batch-of-ONE runs whose bank
    differs by 1e-3 in five rows), not a tie and not the layout. The oracle form of this claim is
    tests/test_gpu_head.py::test_batch_of_independent_streams_vs_oracle_per_stream.
    Criterion: EVERY 3D row (box, score, label) and EVERY 2D row of every (stream, frame) pair has a partner within 1e-3
    (2D boxes in pixels: 0.1) and the 2D counts agree -- except where the pair is accounted for by a tie that is logged
    with its numbers: a ranking cut of the plain run (InstanceBank.update: rank 299 / 300 of the first layer's max-class
    logits, instance_bank.py:137; the decoder's top 300 by score, decoder.py:145) within 1e-4 of flipping; at most 2 pairs."""
    from simpb_amd import configs, plugin
    from simpb_amd.runner import FrameRunner
    wh, bs, frames = (352, 128), 4, 8

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn().half_backbone()

    imgs = [synth.images(bs, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(bs, f, wh, jump=(2, 4, 10.0)) for f in range(frames)]
    model = make()
    batch = FrameRunner(model, bs, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True,
                        independent_streams=True)
    inner, last = model.extract_feat, {}

    def spy(img):
        last["fm"] = inner(img)
        return last["fm"]

    model.extract_feat = spy
    bank = batch.head.instance_bank
    got, seen, states = [], [], []
    for f in range(frames):
        torch.cuda.synchronize()
        states.append({k: v.clone() for k, v in bank._static.items()})
        got.append(batch.step(imgs[f], metas[f]))
        seen.append([t.clone() for t in list(last["fm"])[:3]])
    assert batch.stats["replay"] >= 4 and batch.stats["overflow"] == 0, batch.stats

    def rows3(r):
        b = np.asarray(r["boxes_3d"], np.float64)
        return np.concatenate([b[:, :6], np.sin(b[:, 6:7]), np.cos(b[:, 6:7]), b[:, 7:], np.asarray(r["scores_3d"], np.float64)[:, None],
                               np.asarray(r["labels_3d"], np.float64)[:, None] * 10.0], axis=1)

    def rows2(r):
        return np.concatenate([np.asarray(r["boxes_2d"], np.float64) * 1e-2, np.asarray(r["scores_2d"], np.float64)[:, None],
                               np.asarray(r["labels_2d"], np.float64)[:, None] * 10.0], axis=1)

    full = total = 0
    partial = []
    for b in range(bs):
        replay = _ReplayModel(make().head)
        plain = FrameRunner(replay, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=False)
        mine = plain.head.instance_bank
        seen_outs = {}
        plain.head.register_forward_hook(lambda m, i, o: seen_outs.update(outs=o))
        for f in range(frames):
            if f >= 1:   # the state the batched runner's stream b had in front of this frame
                torch.cuda.synchronize()
                for k, v in states[f].items():
                    mine._static[k].copy_(v if v.dim() == 0 else v[b:b + 1])
            replay.load([seen[f][0][b:b + 1], seen[f][1], seen[f][2]])
            want = plain.step(plain.img, dict(projection_mat=metas[f]["projection_mat"][b:b + 1], image_wh=metas[f]["image_wh"][b:b + 1],
                                              timestamp=metas[f]["timestamp"][b:b + 1], img_metas=[metas[f]["img_metas"][b]]))[0]["img_bbox"]
            have = got[f][b]["img_bbox"]
            assert have["boxes_3d"].shape == want["boxes_3d"].shape
            m3, m2 = _matched_fraction(rows3(have), rows3(want), 1e-3), _matched_fraction(rows2(have), rows2(want), 1e-3)
            ok = m3 == 1.0 and m2 == 1.0 and len(have["boxes_2d"]) == len(want["boxes_2d"])
            if not ok:   # only a ranking cut about to flip accounts for rows without a partner
                o = seen_outs["outs"]
                v = torch.sort(o["classification"][0][0].max(dim=-1).values, descending=True).values
                sc = torch.sort(o["classification"][-1][0].sigmoid().max(dim=-1).values, descending=True).values
                tie = dict(update_cut_gap=float(v[299] - v[300]), decode_cut_gap=float(sc[299] - sc[300]))
                partial.append(dict(stream=b, frame=f, matched_3d=round(m3, 4), matched_2d=round(m2, 4),
                                    n2d=(len(have["boxes_2d"]), len(want["boxes_2d"])), tie=tie))
                assert min(tie.values()) < 1e-4, f"rows without a partner and no ranking cut within 1e-4 of flipping: {partial[-1]}"
                assert m3 >= 0.99 and m2 >= 0.97, partial[-1]   # a flipped cut swaps one instance and its 2D queries, no more
            full += int(ok)
            total += 1
    log_dir = os.environ.get("SIMPB_TEST_LOG_DIR")
    if log_dir and os.path.isdir(log_dir):
        with open(os.path.join(log_dir, "ragged_same_state_ties.json"), "w") as fh:
            json.dump(dict(pairs=total, fully_matched=full, accounted_by_a_tie=partial), fh, indent=1)
    assert len(partial) <= 2, (full, total, partial)


def test_two_pipelined_runners_side_by_side():
    """bench.py --streams: independent camera streams, each its own pipelined runner, launched back to back and
    collected together so that their work (eager warm-up steps and replayed graphs alike) shares the GPU. One stream runs a frame behind the other; each must return what the plain eager runner
    returns for the features it recorded."""
    from simpb_amd import configs, plugin
    from simpb_amd.runner import PipelinedRunner
    wh = (352, 128)

    def make():
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(model)
        return model.cuda().fuse_conv_bn()

    frames = 12
    imgs = [synth.images(1, f % 4, wh).cuda() for f in range(frames)]
    metas = [synth.frame_metas(1, f, wh) for f in range(frames)]
    runners = [PipelinedRunner(make(), 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True)
               for _ in range(2)]
    lag = (0, 1)
    outs, seen = [[], []], [[], []]
    for step in range(frames + 1):
        live = [i for i in range(2) if 0 <= step - lag[i] < frames]
        for i in live:
            runners[i].launch(imgs[step - lag[i]], metas[step - lag[i]])
        for i in live:
            f = step - lag[i]
            outs[i].append(runners[i].collect())
            if f >= 1:
                seen[i].append([t.clone() for t in list(runners[i].fm[(f - 1) % 2])[:3]])
    for i in range(2):
        runners[i].s_bb.synchronize()  # (reading the runner's feature slot from this stream: collect() does not wait for the backbone)
        seen[i].append([t.clone() for t in list(runners[i].fm[(frames - 1) % 2])[:3]])
        out = outs[i][1:] + [runners[i].flush()]
        assert runners[i].stats["replay"] >= 4, runners[i].stats
        _check_against_replay(make().head, seen[i], out, metas, wh, f"stream{i}")


def rows_match_t(a, b, tol):
    from tests.helpers import rows_match
    def wrap(x):  # yaw -> (sin, cos) so that +-pi is not a mismatch
        x = x.double()
        return torch.cat([x[:, :6], x[:, 6:7].sin(), x[:, 6:7].cos(), x[:, 7:]], dim=1).numpy()
    return rows_match(wrap(a), wrap(b), tol)


def test_replayed_frames_of_the_bench_configuration_vs_oracle():
    """What bench.py times, against the oracle: the shipped R50 704x256 configuration -- real fp16 ResNet50 + FPN writing the
    token rows (fp32 + the f16 copy the 2D sampler gathers), the decoder as a replayed hipGraph (split-operand GEMMs and
    attention, fused 3D aggregation, TOK = _Float16 2D sampler, chains, bank / decode kernels) -- frame by frame FROM THE SAME
    STATE: before a replayed frame the bank state is copied out, the frame's tokens are recorded, and OracleHead (CPU, fp32)
    runs the same frame from that state on those tokens. Every detection (3D boxes / scores / labels, 2D boxes, the 2D<->3D
    association, ids up to relabelling: tests.helpers.compare_result, 1e-3) and the bank state the frame leaves (kept rows
    as sets) must agree. Whole-stream comparisons drift (profiles/r04_ragged_cause.md), same-state frames must not."""
    from oracle import simpb_ref as R
    from simpb_amd import configs, plugin
    from simpb_amd.runner import FrameRunner
    from tests.test_gpu_head import _oracle_result_as_golden
    wh = (704, 256)
    cfg = configs.simpb_plus(anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    model = model.cuda().fuse_conv_bn().half_backbone()
    runner = FrameRunner(model, 1, (wh[1], wh[0]), capacity=1536, device=torch.device("cuda"), use_graph=True)
    head = runner.head
    params = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    bank = head.instance_bank
    torch.set_num_threads(16)
    frames, checked = 6, 0
    prev = None
    for f in range(frames):
        metas = synth.frame_metas(1, f, wh)
        img = synth.images(1, f % 4, wh).cuda()
        torch.cuda.synchronize()
        state = {k: v.clone().cpu() for k, v in bank._static.items()}
        got = runner.step(img, metas)[0]["img_bbox"]
        if f >= 3:   # replayed frames (the graph is captured at the second warm frame)
            assert runner.stats["replay"] >= 1
            with torch.no_grad():
                fm = model.extract_feat(img)   # deterministic kernels: the tokens the replayed frame computed from this image
                col = fm[0].float().cpu()
                assert getattr(fm[0], "simpb_f16", None) is not None and torch.equal(fm[0].simpb_f16.float().cpu(), col)
                oracle = R.OracleHead(params, head.operation_order)
                ob = oracle.bank
                ob.cached_feature, ob.cached_anchor = state["cached_feature"].clone(), state["cached_anchor"].clone()
                ob.confidence, ob.instance_id = state["confidence"].clone(), state["instance_id"].clone()
                ob.prev_id = int(state["prev_id"])
                ob.metas = dict(timestamp=prev["timestamp"], img_metas=prev["img_metas"])
                want = oracle.forward([col, fm[1].cpu(), fm[2].cpu()], metas)
                res = oracle.post_process(want, metas)[0]
            compare_result(got, _oracle_result_as_golden(res, "w."), "w.")
            torch.cuda.synchronize()
            for name, w_ in (("cached_feature", ob.cached_feature[0]), ("cached_anchor", ob.cached_anchor[0])):
                a = bank._static[name][0].cpu()
                d = torch.cdist(a.double(), w_.double(), p=float("inf"))
                assert float(torch.maximum(d.min(dim=1).values, d.min(dim=0).values).max()) <= 1e-3, (f, name)
            assert float((torch.sort(bank._static["confidence"][0].cpu()).values - torch.sort(ob.confidence[0]).values).abs().max()) <= 1e-3
            checked += 1
        prev = metas
    assert checked == 3 and runner.stats["overflow"] == 0
