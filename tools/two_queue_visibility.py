"""Stand-alone probe written in round 1 for the eager two-stream fault. It found nothing because the
cause was not visibility: tools/daf_stress.py is the probe that reproduces it (a co-running v_mfma_f32_32x32x16_f16 kernel).

Stream B plays the decoder: every iteration it reads a small persistent buffer with `simpb_bank_get` (identity
ego-motion, zero time step: the output must equal the input), runs some filler kernels and rewrites the buffer with the
next iteration's value from many workgroups; the host waits for both streams at the end of the iteration, like
PipelinedRunner.collect(). Stream A plays the eager backbone: a chain of small convolutions with fresh allocations.
Counts iterations in which the reader returned anything but the current value (its input cloned right in front of it
is counted separately).

    python tools/two_queue_visibility.py [--iters 3000] [--no-backbone]
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--no-backbone", action="store_true")
    args = ap.parse_args()
    from simpb_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda")
    s_a = torch.cuda.Stream(device=dev, priority=0)
    s_b = torch.cuda.Stream(device=dev, priority=-1)
    n = 600
    stored = torch.zeros(1, n, 11, device=dev)
    feat = torch.zeros(1, n, 256, device=dev)
    t_mat = torch.eye(4, device=dev)[None].contiguous()
    dt = torch.zeros(1, device=dev)
    x = torch.randn(6, 64, 32, 88, device=dev)
    ws = [torch.randn(64, 64, 3, 3, device=dev) * 0.05 for _ in range(4)]
    filler = torch.randn(1536, 256, device=dev)
    bad_in = torch.zeros((), device=dev, dtype=torch.int64)
    bad_out = torch.zeros((), device=dev, dtype=torch.int64)
    bad_feat = torch.zeros((), device=dev, dtype=torch.int64)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    torch.cuda.synchronize()
    for it in range(args.iters):
        s_a.wait_stream(torch.cuda.current_stream())
        s_b.wait_stream(torch.cuda.current_stream())
        if not args.no_backbone:
            with torch.cuda.stream(s_a):
                y = x
                for _ in range(10):
                    for w in ws:
                        y = F.relu(F.conv2d(y, w, padding=1)) + 0.1
        with torch.cuda.stream(s_b):
            seen_in = stored.clone()
            warped = torch.empty_like(stored)
            mask = torch.empty(1, dtype=torch.bool, device=dev)
            dt_out = torch.empty(1, device=dev)
            _lib.check(lib.simpb_bank_get(ptr(warped), ptr(mask), ptr(dt_out), ptr(stored), ptr(t_mat), ptr(dt), 1, n, 2.0, 0.5,
                                          ctypes.c_void_p(s_b.cuda_stream)), "simpb_bank_get")
            seen_out = warped.clone()
            seen_feat = feat.clone()
            bad_in += (seen_in != float(it)).any()
            bad_out += (seen_out[..., :6] != float(it)).any()
            bad_feat += (seen_feat != float(it)).any()
            z = filler
            for _ in range(40):  # decoder-like filler: small dependent kernels with fresh allocations
                z = torch.tanh(z * 1.01 + 0.01)
            # next value, written from many workgroups (row gather like bank_gather)
            src_a = torch.full((1, 900, 11), float(it + 1), device=dev)
            src_f = torch.full((1, 900, 256), float(it + 1), device=dev)
            idx = torch.arange(n, device=dev)
            stored.copy_(src_a.index_select(1, idx))
            feat.copy_(src_f.index_select(1, idx))
        s_b.synchronize()
        s_a.synchronize()
        if it % 500 == 499:
            print(f"iter {it + 1}: stale reader inputs {int(bad_in)}, stale reader outputs {int(bad_out)}, "
                  f"stale feature reads {int(bad_feat)}", flush=True)
    print(f"RESULT iters={args.iters} backbone={'off' if args.no_backbone else 'on'} bad_in={int(bad_in)} "
          f"bad_out={int(bad_out)} bad_feat={int(bad_feat)}", flush=True)


if __name__ == "__main__":
    main()
