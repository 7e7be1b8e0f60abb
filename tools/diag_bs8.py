import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpb_amd.runner import FrameRunner
sys.argv += ["--bs", "8", "--capacity", "4096"]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev)
imgs = bench.make_frames(args, dev, 8)
r = FrameRunner(model, args.bs, (256, 704), capacity=4096, device=dev, use_graph=False)
for f in range(5):
    r.step(imgs[f % 4], bench.frame_metas(args, f))
    for i in (0, 17, 34):
        a = model.head.layers[i].last
        print("frame", f, "layer", i, "count per stream", a.count.sum(1).tolist(), "group_start", a.group_start.tolist())
