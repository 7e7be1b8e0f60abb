"""Which kernel MIOpen picks for one fp16 channels_last convolution on this box (run under rocprofv3 --kernel-trace --stats).
usage: python3 tools/conv_kernel_name.py cin cout h w k stride"""
import sys

import torch
import torch.nn.functional as F

cin, cout, h, w, k, stride = (int(v) for v in sys.argv[1:7])
torch.backends.cudnn.benchmark = True
x = torch.randn(6, cin, h, w, device="cuda").half().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(cout, cin, k, k, device="cuda") * 0.02).half().contiguous(memory_format=torch.channels_last)
for _ in range(3):
    F.conv2d(x, wt, None, stride=stride, padding=k // 2)
torch.cuda.synchronize()
for _ in range(200):
    F.conv2d(x, wt, None, stride=stride, padding=k // 2)
torch.cuda.synchronize()
