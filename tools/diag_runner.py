"""Where does a FrameRunner step spend its wall time? (host phases vs GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpb_amd.runner import FrameRunner
from simpb_amd.plugin.detection3d import SparseBox3DDecoder
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev)
imgs = bench.make_frames(args, dev, 8)
metas = [bench.frame_metas(args, f) for f in range(40)]
r = FrameRunner(model, args.bs, (args.image_wh[1], args.image_wh[0]), capacity=args.capacity, device=dev, use_graph=not args.eager)
for f in range(6):
    r.step(imgs[f % 4], metas[f])
torch.cuda.synchronize()
acc = dict(stage=0, launch=0, gpu_wait=0, readback=0, host_decode=0)
n = 20
for f in range(6, 6 + n):
    t0 = time.perf_counter()
    r._stage(imgs[f % 4], metas[f]); dm = r._device_metas(metas[f])
    t1 = time.perf_counter()
    if r.graph is not None:
        r.graph.replay(); rec = r.outputs
    else:
        rec = r._frame(dm, metas[f]["img_metas"][0]["aug_config"])
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    h3, h2, ov = r._read_back(*rec)
    t4 = time.perf_counter()
    SparseBox3DDecoder.decode_static_host(h3, h2, 6)
    t5 = time.perf_counter()
    r.prev_metas = dict(img_metas=metas[f]["img_metas"])
    print(f'frame {f}: launch {1e3*(t2-t1):.2f} gpu_wait {1e3*(t3-t2):.2f}', flush=True)
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[k] += v
print({k: round(v / n * 1e3, 3) for k, v in acc.items()}, "ms per frame; graph =", r.graph is not None)
