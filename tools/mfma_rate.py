"""Issue rate of the matrix instructions this code base may use, measured with tools/mfma_burn.hip (register operands only):
1024 workgroups x 4 waves, every SIMD of the chip busy with one dependent chain per wave. Prints cycles per instruction
at the clock the chip holds and TFLOP/s. usage: python tools/mfma_rate.py"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(HERE, "..", "gpurun_out", "libmfma_burn.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "mfma_burn.hip"), "-o", out], check=True)
lib = ctypes.CDLL(out)
sink = torch.zeros(16, device="cuda")
NAMES = {0: ("32x32x16_f16", 2 * 32 * 32 * 16), 1: ("16x16x32_f16", 2 * 16 * 16 * 32), 4: ("32x32x8_f16", 2 * 32 * 32 * 8),
         6: ("16x16x16_f16", 2 * 16 * 16 * 16), 5: ("32x32x2_f32", 2 * 32 * 32 * 2),
         # four independent accumulators per loop trip (flop per trip = 4 instructions)
         7: ("4 x 4x4x1_f32", 4 * 2 * 16 * 4 * 4 * 1), 8: ("4 x 4x4x4_f16", 4 * 2 * 16 * 4 * 4 * 4), 9: ("4 x 16x16x4_f32", 4 * 2 * 16 * 16 * 4),
         10: ("4 x 16x16x16_f16", 4 * 2 * 16 * 16 * 16)}
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for v, (name, flop) in NAMES.items():
    for blocks in (256, 1024):   # one / four waves per SIMD
        iters = 20000
        lib.mfma_burn(v, ctypes.c_void_p(sink.data_ptr()), blocks, 100, stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.mfma_burn(v, ctypes.c_void_p(sink.data_ptr()), blocks, iters, stream)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        waves_per_simd = blocks * 4 / 1024
        total = blocks * 4 * iters * flop
        print(f"{name:14s} blocks {blocks:5d}: {ms:7.3f} ms  {total / ms * 1e-9:7.1f} TFLOP/s  "
              f"{ms * 1e-3 / (iters * waves_per_simd) * 1e9:6.2f} ns per instruction per SIMD", flush=True)
