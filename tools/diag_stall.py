"""Is the occasional ~50 ms stall tied to graph replay, to host work between replays, or to the box?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpb_amd.runner import FrameRunner
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev)
imgs = bench.make_frames(args, dev, 8)
metas = [bench.frame_metas(args, f) for f in range(12)]
r = FrameRunner(model, args.bs, (args.image_wh[1], args.image_wh[0]), capacity=args.capacity, device=dev)
for f in range(6):
    r.step(imgs[f % 4], metas[f])
torch.cuda.synchronize()
def series(name, fn, n=60):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    ts_s = sorted(ts)
    print(f"{name}: median {ts_s[n//2]:.2f} ms, p90 {ts_s[int(n*0.9)]:.2f}, max {ts_s[-1]:.2f}, mean {sum(ts)/n:.2f}; >2x median: {sum(t > 2*ts_s[n//2] for t in ts)}", flush=True)
    return ts

from simpb_amd.plugin.detection3d import SparseBox3DDecoder
def with_stage():
    r._stage(imgs[1], metas[7]); r.graph.replay()
def with_readback():
    r.graph.replay(); r._read_back(*r.outputs)
def with_decode():
    r.graph.replay(); h3, h2, _ = r._read_back(*r.outputs); SparseBox3DDecoder.decode_static_host(h3.numpy(), h2.numpy(), 6)
def full():
    r._stage(imgs[1], metas[7]); r.graph.replay(); h3, h2, _ = r._read_back(*r.outputs); SparseBox3DDecoder.decode_static_host(h3.numpy(), h2.numpy(), 6)
series("A replay only", r.graph.replay)
series("E stage+replay", with_stage)
series("F replay+readback", with_readback)
series("G replay+readback+decode", with_decode)
series("H full", full)
import gc; gc.disable()
series("H' full, gc disabled", full)
gc.enable()
