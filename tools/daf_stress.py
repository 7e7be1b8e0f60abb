"""Stand-alone stress of the 3D deformable-aggregation kernel beside a busy second stream (round 2: the kernel gives a wrong
partial sum in channels 192-255 of an anchor now and then when another hardware queue keeps the chip busy).

One stream launches daf_fwd_rows over and over on fixed operands of the shipped shapes and compares each output with the
output of the same launch on an idle device; a second stream loops a co-runner kernel. Reports faulty launches, rows and
the channel blocks that differ.

    python tools/daf_stress.py [--k16] [--co conv1x1|linear_split|conv3x3|linear_f32|matmul|copy|bias_act|format|none]
                               [--launches N] [--cu-split K]
--k16 rebuilds the library (into gpurun_out/k16/) with the single-instruction FP16 matrix step v_mfma_f32_32x32x16_f16
(-DSIMPB_MFMA_F16_K16=1, csrc/mfma_f16.h) instead of the product's two v_mfma_f32_32x32x8f16 steps, and runs with that.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k16", action="store_true")
    ap.add_argument("--co", default="conv1x1")
    ap.add_argument("--launches", type=int, default=3000)
    ap.add_argument("--cu-split", type=int, default=0)
    ap.add_argument("--anchors", type=int, default=900)
    ap.add_argument("--victim", default="daf",
                    help="the kernel that is re-launched and compared bit for bit with its idle-device output: daf (default), or one "
                         "of the decoder's dense kernels that keep packed-FP32 instructions: layernorm | attention | chain | gemm")
    ap.add_argument("--build-flags", default="",
                    help="extra hipcc flags for a variant build of the whole library (e.g. -fno-slp-vectorize: no packed-FP32 "
                         "instructions in the victim kernel), combined with --k16 when both are given")
    args = ap.parse_args()
    import simpb_amd._lib as L
    if args.k16 or args.build_flags:
        from simpb_amd import build
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out", "k16"), exist_ok=True)
        flags = (["-DSIMPB_MFMA_F16_K16=1"] if args.k16 else []) + args.build_flags.split()
        L.LIB = build.build_extension(extra_flags=flags, out=os.path.join(root, "gpurun_out", "k16", "libsimpb_hip_variant.so"))
        print("variant build:", flags, flush=True)
    from simpb_amd.plugin import ops
    from _streams import cu_masked_streams
    dev = torch.device("cuda")
    L.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = [(64, 176), (32, 88), (16, 44), (8, 22)]
    maps = [torch.randn(1, 6, 256, h, w, generator=g).cuda() for h, w in shapes]
    col, ss, ssi = ops.feature_maps_format(maps)
    A, P, K, Lv, G = args.anchors, 13, 6, 4, 8
    loc = (torch.rand(1, A, P, K, 2, generator=g) * 1.6 - 0.3).cuda()
    w = torch.softmax(torch.randn(1, A, P * K * Lv, G, generator=g), dim=2).reshape(1, A, P, K, Lv, G).permute(0, 1, 2, 3, 4, 5).contiguous().cuda()
    ss32, ssi32 = ss.int().contiguous(), ssi.int().contiguous()

    def daf():
        return ops.deformable_aggregation_function(col, ss32, ssi32, loc, w)

    if args.victim != "daf":
        # the decoder's dense kernels whose files are NOT built with NO_PACKED_FP32 (simpb_amd/build.py): row statistics of the
        # LayerNorm (csrc/gemm.hip), the softmax of the attention kernels, the chain kernel's LayerNorm / post stages, the GEMM epilogue
        from simpb_amd.plugin import dense
        victim_x = torch.randn(1, 900, 512, generator=g).cuda()   # (its own name: `vx` below is a co-runner operand)
        if args.victim == "layernorm":
            ln = torch.nn.LayerNorm(512).cuda()
            daf = lambda: dense.layernorm([victim_x[..., :256], victim_x[..., 256:]], ln)   # noqa: E731
        elif args.victim == "attention":
            qkv = torch.randn(1, 900, 1536, generator=g).cuda()
            daf = lambda: ops.attention_f32(qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:], 8, split=1)   # noqa: E731
        elif args.victim == "gemm":
            gw_ = (torch.randn(256, 512, generator=g) / 22).cuda()
            gb_ = torch.randn(256, generator=g).cuda()
            daf = lambda: dense.linear(victim_x, gw_, gb_, relu=True)   # noqa: E731
        elif args.victim == "chain":
            from simpb_amd.plugin.detection3d import SparseBox3DEncoder
            from simpb_amd import synth
            enc = SparseBox3DEncoder(embed_dims=[128, 32, 32, 64], vel_dims=3, mode="cat", output_fc=False, in_loops=1, out_loops=4)
            synth.load_procedural(enc, seed=3)
            enc = enc.cuda()
            box = torch.randn(1, 900, 11, generator=g).cuda()
            daf = lambda: enc(box)   # noqa: E731
        else:
            raise SystemExit("unknown --victim")

    ref = daf()
    torch.cuda.synchronize()
    for _ in range(3):
        assert torch.equal(daf(), ref), "not deterministic on an idle device"
    if args.cu_split:
        s_co, s_daf = cu_masked_streams(dev, args.cu_split)
    else:
        s_co, s_daf = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    import torch.nn.functional as F
    cx = torch.randn(6, 256, 64, 176, device=dev).half().contiguous(memory_format=torch.channels_last)
    cw3 = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).half().contiguous(memory_format=torch.channels_last)
    cw1 = (torch.randn(256, 256, 1, 1, device=dev) * 0.02).half()
    cb = torch.zeros(256, device=dev).half()
    big = torch.randn(23 * 1024 * 1024, device=dev)
    mm = torch.randn(4096, 4096, device=dev)
    vx = torch.randn(6, 14960, 256, device=dev)
    vw = torch.randn(256, 256, device=dev) * 0.05
    vb = torch.zeros(256, device=dev)
    lv = [torch.randn(6, 256, h, w, device=dev).half().contiguous(memory_format=torch.channels_last) for h, w in shapes]

    BURN = {"f16_32x32x16": 0, "f16_16x16x32": 1, "bf16_32x32x16": 2, "bf16_16x16x32": 3, "f16_32x32x8": 4, "f32_32x32x2": 5,
            "f16_16x16x16": 6}
    burn = None
    if args.co.startswith("burn:"):
        # a co-runner that only issues one matrix instruction in a loop on registers (tools/mfma_burn.hip, built here)
        import ctypes
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(os.path.dirname(here), "gpurun_out", "libmfma_burn.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                        os.path.join(here, "mfma_burn.hip")], check=True, stderr=subprocess.DEVNULL)
        blib = ctypes.CDLL(so)
        blib.mfma_burn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        sink = torch.zeros(4, device=dev)

        def burn(variant):
            rc = blib.mfma_burn(variant, sink.data_ptr(), 2048, 600, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

    gx = gw = None
    gstride = gk = 1
    if args.co.startswith("conv:"):  # conv:cin,cout,h,w,k,stride  (fp16 channels_last, 6 images, as MIOpen picks the kernel)
        cin, cout, h, w_, gk, gstride = (int(v) for v in args.co[5:].split(","))
        torch.backends.cudnn.benchmark = True
        gx = torch.randn(6, cin, h, w_, device=dev).half().contiguous(memory_format=torch.channels_last)
        gw = (torch.randn(cout, cin, gk, gk, device=dev) * 0.02).half().contiguous(memory_format=torch.channels_last)
        for _ in range(3):
            F.conv2d(gx, gw, None, stride=gstride, padding=gk // 2)
        torch.cuda.synchronize()
    bb_model = bb_img = None
    if args.co == "backbone":
        # the shipped backbone + FPN + token format + value projections exactly as the runner's backbone stream runs them
        # (vendor 3x3 / 7x7 convolutions as MIOpen picks them on this box, our conv1x1 / value_proj): the whole co-runner
        from simpb_amd import configs, plugin, synth
        cfg = configs.simpb_plus(anchor=synth.anchors(900))
        bb_model = plugin.build_detector(cfg["model"]).eval()
        synth.load_procedural(bb_model)
        bb_model = bb_model.cuda().fuse_conv_bn().half_backbone()
        bb_img = synth.images(1, 0, (704, 256)).cuda()
        torch.backends.cudnn.benchmark = True
        with torch.no_grad():
            for _ in range(3):
                fm = bb_model.extract_feat(bb_img)
                bb_model.head.precompute_values(list(fm))
        torch.cuda.synchronize()

    def co():
        if args.co == "conv1x1":
            ops.conv1x1_nhwc(cx, cw1, cb, None, True, 1)
        elif args.co == "conv3x3":
            F.conv2d(cx, cw3, None, padding=1)
        elif args.co == "copy":
            big.clone()
        elif args.co == "matmul":
            mm @ mm
        elif args.co == "linear_split":
            ops.linear_split(vx, vw, vb)
        elif args.co == "linear_f32":
            ops.linear_f32(vx, vw, vb)
        elif args.co == "bias_act":
            ops.bias_act_(cx, cb, None, True)
        elif args.co == "format":
            ops.format_tokens(lv, 1, 6)
        elif args.co.startswith("burn:"):
            burn(BURN[args.co[5:]])
        elif args.co.startswith("conv:"):
            F.conv2d(gx, gw, None, stride=gstride, padding=gk // 2)
        elif args.co == "maxpool":
            F.max_pool2d(cx, 3, 2, 1)
        elif args.co == "backbone":
            with torch.no_grad():
                bb_model.extract_feat(bb_img)
        elif args.co != "none":
            raise SystemExit("unknown --co")

    bad_launch = torch.zeros((), dtype=torch.long, device=dev)
    bad_rows = torch.zeros((), dtype=torch.long, device=dev)
    blocks = torch.zeros(4, dtype=torch.long, device=dev)
    torch.cuda.synchronize()
    for it in range(args.launches):
        with torch.cuda.stream(s_co):
            for _ in range(1 if args.co == "backbone" else 3):
                co()
        with torch.cuda.stream(s_daf):
            out = daf()
            d = (out != ref)
            rows = d.any(-1).sum()
            bad_rows += rows
            bad_launch += (rows > 0).long()
            if d.shape[-1] % 256 == 0:
                blocks += d.reshape(-1, 4, d.shape[-1] // 4).any(-1).sum(0)
        if it % 200 == 199:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"victim={args.victim} co={args.co} fp16_matrix_step={'1 x 32x32x16' if args.k16 else '2 x 32x32x8'} cu_split={args.cu_split}: {int(bad_launch)} faulty launches of {args.launches}, "
          f"{int(bad_rows)} rows; faulty rows by channel block [0-63, 64-127, 128-191, 192-255]: {blocks.tolist()}", flush=True)


if __name__ == "__main__":
    main()
