import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
args = bench.parse()
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
model = bench.build_model(args, dev)
img = bench.make_frames(args, dev, 1)[0]
with torch.no_grad():
    for _ in range(5):
        model.extract_feat(img)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.extract_feat(img)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
