"""Stage timing of one bench step (progress lines to stdout)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:7.1f}s]", *a, flush=True)
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev); log("model built")
imgs = bench.make_frames(args, dev, 8); log("frames staged")
torch.cuda.synchronize()
with torch.no_grad():
    for f in range(6):
        m = bench.frame_metas(args, dev, f)
        torch.cuda.synchronize(); a = time.time()
        fm = model.extract_feat(imgs[f % 4]); torch.cuda.synchronize(); b = time.time()
        outs = model.head(fm, m); torch.cuda.synchronize(); c = time.time()
        res = model.head.post_process(outs, m); torch.cuda.synchronize(); d = time.time()
        log(f"frame {f}: backbone+fpn+format {1e3*(b-a):.1f} ms, head {1e3*(c-b):.1f} ms, post {1e3*(d-c):.1f} ms")
