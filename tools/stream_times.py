"""What bounds a pipelined frame: the replayed backbone graph alone, the replayed decoder graph alone, and both side by
side (the steady state of runner.PipelinedRunner), timed with events on their streams. usage: python tools/stream_times.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpb_amd import configs, plugin, synth  # noqa: E402
from simpb_amd.runner import PipelinedRunner  # noqa: E402

# --route name=0/1 [...]: take the other branch of a route switch for this measurement (simpb_amd/plugin/routes.py)
import contextlib  # noqa: E402
from simpb_amd.plugin import routes  # noqa: E402
_stack = contextlib.ExitStack()
while "--route" in sys.argv:
    i = sys.argv.index("--route")
    k, v = sys.argv[i + 1].split("=")
    _stack.enter_context(routes.override(**{k: bool(int(v))}))
    del sys.argv[i:i + 2]
print("routes:", routes.R, flush=True)
BS = 1
if "--bs" in sys.argv:   # --bs N: N camera streams through one batched runner
    i = sys.argv.index("--bs")
    BS = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
wh = (704, 256)
cfg = configs.simpb_plus(anchor=synth.anchors(900))
model = plugin.build_detector(cfg["model"]).eval()
synth.load_procedural(model)
model = model.cuda().fuse_conv_bn().half_backbone()
torch.backends.cudnn.benchmark = True
if "--prio" in sys.argv:   # --prio BB HEAD: stream priorities of the backbone / decoder streams (default 0 -1: decoder first)
    i = sys.argv.index("--prio")
    PipelinedRunner.STREAM_PRIORITIES = (int(sys.argv[i + 1]), int(sys.argv[i + 2]))
    print("stream priorities (backbone, decoder):", PipelinedRunner.STREAM_PRIORITIES, torch.cuda.Stream.priority_range(), flush=True)
REF_BATCH = "--reference-batch" in sys.argv   # bs > 1 with the reference's padded camera groups instead of independent streams
if REF_BATCH:
    sys.argv.remove("--reference-batch")
r = PipelinedRunner(model, BS, (wh[1], wh[0]), capacity=1536 if BS == 1 or not REF_BATCH else 2048, device=torch.device("cuda"),
                    independent_streams=not REF_BATCH)
XCD_DROP = 0
if "--bb-xcd-drop" in sys.argv:   # --bb-xcd-drop K: the backbone stream loses K whole XCDs (CU mask bit i belongs to XCD i % 8); the decoder keeps the chip
    i = sys.argv.index("--bb-xcd-drop")
    XCD_DROP = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
DEC_ONLY = "--dec-only" in sys.argv   # for rocprofv3 --kernel-trace: decoder graph replays with nothing beside them
if len(sys.argv) > 1 and not DEC_ONLY and "--bb-only" not in sys.argv and "--co" not in sys.argv and "--prio" not in sys.argv:
    # --bb-drop N: the backbone stream loses one group of 8 CUs in every N groups (in every XCD, whichever way CU indices
    # map to XCDs); the decoder stream keeps the whole chip
    import ctypes
    drop = int(sys.argv[1])
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    m = (ctypes.c_uint32 * words)()
    kept = 0
    for i in range(n_cu):
        if (i // 8) % drop != 0:
            m[i // 32] |= 1 << (i % 32)
            kept += 1
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), m) == 0
    r.s_bb = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))
    print(f"backbone stream on {kept} of {n_cu} CUs", flush=True)
if XCD_DROP:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    m = (ctypes.c_uint32 * words)()
    kept = 0
    for i in range(n_cu):
        if i % 8 >= XCD_DROP:
            m[i // 32] |= 1 << (i % 32)
            kept += 1
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), m) == 0
    r.s_bb = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))
    print(f"backbone stream on {kept} of {n_cu} CUs ({XCD_DROP} XCD(s) left to the decoder alone)", flush=True)
imgs = [synth.images(BS, f, wh).cuda() for f in range(4)]
metas = [synth.frame_metas(BS, f, wh) for f in range(60)]
for f in range(24):
    r.step(imgs[f % 4], metas[f])
torch.cuda.synchronize()
assert all(g is not None for g in r.bb_graph) and all(g is not None for g in r.head_graph), "graphs not captured yet"
n = 50


def wall(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def bb():
    with torch.cuda.stream(r.s_bb):
        for i in range(n):
            r.bb_graph[i & 1].replay()


def dec():
    with torch.cuda.stream(r.s_head):
        for i in range(n):
            r.head_graph[i & 1].replay()


def both():
    for i in range(n):
        with torch.cuda.stream(r.s_bb):
            r.bb_graph[i & 1].replay()
        with torch.cuda.stream(r.s_head):
            r.head_graph[i & 1].replay()


if "--co" in sys.argv:
    # the decoder graph beside ONE kind of backbone kernel looping on the other stream: which ingredient of the backbone
    # slows the decoder (matrix load, memory traffic, nothing in particular)?
    from simpb_amd.plugin import ops
    x256 = torch.randn(6, 256, 64, 176, device="cuda").half().contiguous(memory_format=torch.channels_last)
    x64 = torch.randn(6, 64, 64, 176, device="cuda").half().contiguous(memory_format=torch.channels_last)
    w3 = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).half().contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(256, 64, 1, 1, device="cuda") * 0.05).half()
    b256 = torch.zeros(256, device="cuda").half()
    res = torch.randn(6, 256, 64, 176, device="cuda").half().contiguous(memory_format=torch.channels_last)
    tok16 = torch.randn(89760, 256, device="cuda").half()
    vw, vb = torch.randn(256, 256, device="cuda") * 0.05, torch.zeros(256, device="cuda")
    big = torch.randn(46 * 1024 * 1024, device="cuda")
    big2 = torch.empty_like(big)
    co = {"conv3x3": (lambda: ops.conv3x3_nhwc(x256, w3, b256, True, 1), 145.0),
          "conv1x1": (lambda: ops.conv1x1_nhwc(x64, w1, b256, res, True, 1), 25.0),
          "value_proj": (lambda: ops.linear_split(tok16, vw, vb), 75.0),
          "copy": (lambda: big2.copy_(big), 75.0)}
    m = 20
    for name, (fn, us) in co.items():
        reps = int(m * 2600 / us * 1.3)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(r.s_bb):
            for _ in range(reps):
                fn()
        with torch.cuda.stream(r.s_head):
            a.record()
            for i in range(m):
                r.head_graph[i & 1].replay()
            b.record()
        torch.cuda.synchronize()
        print(f"decoder graph beside a loop of {name:10s}: {a.elapsed_time(b) / m:.3f} ms per frame", flush=True)
    sys.exit(0)
if DEC_ONLY:
    dec()
    torch.cuda.synchronize()
    sys.exit(0)
if "--bb-only" in sys.argv:   # for rocprofv3 --kernel-trace: backbone graph replays with nothing beside them
    bb()
    torch.cuda.synchronize()
    sys.exit(0)
for name, fn in (("backbone graph alone", bb), ("decoder graph alone", dec), ("both streams side by side", both)):
    fn()
    print(f"{name:28s} {wall(fn):.3f} ms per step of {BS} frame(s)", flush=True)
