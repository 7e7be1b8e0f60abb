"""1x1 convolutions of the fp16 channels_last backbone as vendor GEMMs with a fused bias(+ReLU) epilogue
(torch._addmm_activation) against F.conv2d + the in-place bias_act kernel. usage: python tools/bench_conv1x1_gemm.py"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from simpb_amd.plugin.ops import bias_act_  # noqa: E402
from tools.bench_conv_fused import timeit  # noqa: E402

torch.backends.cudnn.benchmark = True
SHAPES = [(64, 64, 64, 176), (64, 256, 64, 176), (256, 64, 64, 176), (256, 128, 64, 176), (128, 512, 32, 88), (512, 128, 32, 88),
          (512, 256, 32, 88), (256, 1024, 16, 44), (1024, 256, 16, 44), (1024, 512, 16, 44), (512, 2048, 8, 22), (2048, 512, 8, 22)]
for cin, cout, h, w in SHAPES:
    x = torch.randn(6, cin, h, w, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 1, 1, device="cuda", dtype=torch.half).contiguous(memory_format=torch.channels_last) * 0.05
    b = torch.randn(cout, device="cuda", dtype=torch.half)
    w2 = wt.reshape(cout, cin).t().contiguous()  # [cin, cout]
    x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
    assert x2.data_ptr() == x.data_ptr()
    ours = lambda: bias_act_(F.conv2d(x, wt, None, 1, 0), b, None, relu=True)  # noqa: E731
    gemm = lambda: torch._addmm_activation(b, x2, w2, use_gelu=False)  # noqa: E731
    gemm_nt = lambda: torch._addmm_activation(b, x2, wt.reshape(cout, cin).t(), use_gelu=False)  # noqa: E731
    ref = ours().permute(0, 2, 3, 1).reshape(-1, cout)
    err = float((ref.float() - gemm().float()).abs().max())
    print(f"{cin:4d}->{cout:4d} {h}x{w}: conv+bias_act {timeit(ours):6.1f} us   addmm_act (W^T copy) {timeit(gemm):6.1f} us   "
          f"addmm_act (W view) {timeit(gemm_nt):6.1f} us   max diff {err:.3g}")
