"""Diagnostic: per-phase cycle stamps of one workgroup of the MLP-chain kernel (refine3d reg chain).
Builds a separate library with -DSIMPB_CHAIN_STAMPS into /tmp; the product library is not touched.
usage: python tools/chain_stamps.py"""
import ctypes
import glob
import os
import subprocess
import sys

import torch

sys.path.insert(0, ".")
from simpb_amd import _lib, build  # noqa: E402

dbg = "/tmp/libsimpb_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSIMPB_CHAIN_STAMPS",
                "-o", dbg] + build.sources(), check=True, stderr=subprocess.DEVNULL)
build.LIB = dbg
_lib.LIB = dbg
from simpb_amd.plugin.detection3d import SparseBox3DEncoder, SparseBox3DRefinementModule  # noqa: E402

torch.manual_seed(0)
r3 = SparseBox3DRefinementModule(256, num_cls=10, refine_yaw=True, with_quality_estimation=True).cuda().eval()
enc3 = SparseBox3DEncoder([128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4).cuda().eval()
a3 = torch.randn(1, 900, 11, device="cuda")
f3, e3 = torch.randn(1, 900, 256, device="cuda"), torch.randn(1, 900, 256, device="cuda")
dt = torch.tensor([0.5], device="cuda")
lib = _lib.lib()
lib.simpb_debug_chain_stamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 128)()
for name, fn, nops in [("refine3d reg", lambda: r3(f3, a3, e3, time_interval=dt, return_cls=False), 7),
                       ("anchor encoder (pos branch)", lambda: enc3(a3), 8)]:
    with torch.no_grad():
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    lib.simpb_debug_chain_stamps(buf)
    t = list(buf)
    print(name, "total cycles", t[127] - t[0], "input", t[1] - t[0], "output", t[127] - t[126])
    for o in range(nops):
        b = 2 + 4 * o
        print(f"  op {o}: start->first weights {t[b + 1] - t[b]:6d}  matrix loop+epilogue {t[b + 2] - t[b + 1]:6d}  "
              f"barrier {t[b + 3] - t[b + 2]:6d}   (whole op {t[b + 3] - t[b]:6d})")
