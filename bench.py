"""Headline benchmark: frames/sec of SimPB+ (ResNet50 704x256, 6 cameras) on MI355X, one process
per GPU. A step = one 6-camera frame per stream through backbone+FPN (PyTorch-ROCm, fp16) and the
decoder hot path (HIP kernels, fp32) including post_process. Inputs are synthetic and resident in
HBM before the timed region; weights are procedural random-init of the shipped architecture.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job frames/s,
`roofline` for the 3D deformable-aggregation kernel (HIP events on the launch stream around each of its
launches in a few instrumented frames run right after the timed region -- event nodes cannot be read back
from inside a replayed graph; algorithmic bytes per SURVEY.md §8d), `cpu_baseline` = the oracle (CPU restatement
of the reference) timed on the host cores for a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node, one rank each (default: WORLD_SIZE under a launcher, otherwise 1)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prime", type=int, default=10,
                    help="untimed set-up frames before the warm-up: MIOpen conv search, lazy initialisation and the "
                         "hipGraph captures happen here (the model-compilation step of this path)")
    ap.add_argument("--bs", type=int, default=1, help="camera streams per GPU (BASELINE config #2: 1)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--image-wh", type=int, nargs=2, default=(704, 256))
    ap.add_argument("--capacity", type=int, default=1536, help="static 2D query slots (N2 is ~1.1-1.2k at R50 704x256)")
    ap.add_argument("--eager", action="store_true", help="do not replay the frame as a hipGraph")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap backbone(t+1) with decoder(t) (simpb_amd.runner.PipelinedRunner)")
    ap.add_argument("--no-split", action="store_true",
                    help="keep the single-frame decoder layer of frame t+1 behind the temporal part of frame t "
                         "(simpb_amd.runner.PipelinedRunner instead of SplitPipelinedRunner)")
    ap.add_argument("--meter-frames", type=int, default=8, help="instrumented eager frames for the roofline leg")
    ap.add_argument("--no-conv-search", action="store_true",
                    help="do not let MIOpen benchmark convolution algorithms during warm-up (cudnn.benchmark off)")
    ap.add_argument("--route", action="append", default=[], metavar="NAME=0|1",
                    help="measurement only: take the other branch of a route switch (simpb_amd/plugin/routes.py) for this run")
    ap.add_argument("--reference-batch", action="store_true",
                    help="with --bs > 1: the reference's batch semantics (camera groups padded to the max over the batch, "
                         "allocation.py:91-99) instead of bs independent streams decoded as batches of one")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent camera streams per GPU, each a bs-sized runner of its own replayed concurrently "
                         "(BASELINE config #3 shape: 8 streams per GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--h2d", action="store_true",
                    help="frames start in pinned host memory and cross PCIe inside the timed step (async copy on the backbone "
                         "stream, beside the previous frame's decoder); default: inputs resident in HBM, as `value` requires")
    ap.add_argument("--token-std", type=float, default=None,
                    help="synthetic-weight conditioning for configurations other than the default: rescale the FPN's output "
                         "convolutions (linear in their weights and biases) so that the camera tokens have this standard deviation. "
                         "A random-weight ResNet101 leaves tokens an order of magnitude larger than the ResNet50's, which drives the "
                         "refined anchors out of view (V = 2 252 valid triples instead of ~12.9 k): with the R50's token scale "
                         "(reported as config.token_std) the decoder sits at the operating point SURVEY.md 8d describes")
    ap.add_argument("--residual-damp", type=float, default=1.0,
                    help="synthetic-weight conditioning for deep backbones: scale the last BatchNorm (gain and bias) of every "
                         "bottleneck's residual branch. He-initialised random weights let the activations of a ResNet101 grow by a "
                         "factor per block until the fp16 maps overflow (tokens NaN at 33 blocks; round 3's R101 line ran on those); "
                         "the default workload (ResNet50, 16 blocks) stays as it is")
    ap.add_argument("--h2d-steps", type=int, default=30,
                    help="steps of the secondary leg that repeats the measurement with the frames crossing PCIe (the reference's "
                         "protocol); reported as reference_protocol_h2d, never as value; 0 = skip")
    ap.add_argument("--capacity-by-rank", type=int, nargs="+", default=None, metavar="SLOTS",
                    help="rehearsal only: the static 2D capacity of rank r (so that ONE rank overflows, re-runs and re-captures "
                         "while the others keep submitting: the exchange must stay matched)")
    ap.add_argument("--lib", default=None,
                    help="measurement: load this build of the C-ABI library instead of the in-tree one (a variant built with "
                         "other compiler flags: python -c 'from simpb_amd import build; build.build_extension(extra_flags=[...], out=PATH)')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=21, help="frames of the CPU baseline leg: 1 cold + the warm frames that are timed")
    return ap.parse_args()


class KernelMeter:
    """Roofline leg. Durations come from HIP event pairs the library records on the launch stream
    directly around each sampler launch (simpb_timing_*, include/simpb_hip.h), for every sampler launch of
    the instrumented frames that follow the timed region. The wrappers here only remember each launch's shapes (and the DAF sampling
    locations, to count valid triples afterwards) so algorithmic bytes can be computed."""

    DAF, MSDA = 1, 2

    def __init__(self, steps):
        from simpb_amd import _lib
        from simpb_amd.plugin import blocks, group_attn
        self.lib = _lib.lib()
        self.blocks, self.group_attn = blocks, group_attn
        self.daf_orig, self.msda_orig, self.fused_orig = blocks.DAF, group_attn.ms_deform_attn_grouped, blocks.dfa_fused
        self.daf_calls, self.msda_calls = [], []
        self.enabled = False
        self.capacity = 8 * steps + 64
        _lib.check(self.lib.simpb_timing_enable(self.capacity), "simpb_timing_enable")

    def __enter__(self):
        def daf(feat, ss, ssi, loc, w):
            out = self.daf_orig(feat, ss, ssi, loc, w)
            if self.enabled:
                self.daf_calls.append((loc, tuple(w.shape), tuple(out.shape)))
            return out

        def msda(value, ss, lsi, loc, aw, qcam):
            out = self.msda_orig(value, ss, lsi, loc, aw, qcam)
            if self.enabled:
                self.msda_calls.append((tuple(loc.shape), value.shape[-1], qcam))
            return out

        def fused(feat, ss, ssi, anchor, learn, fix_scale, proj, wh, feat_logits, cam_logits, groups, **kw):
            out = self.fused_orig(feat, ss, ssi, anchor, learn, fix_scale, proj, wh, feat_logits, cam_logits, groups, **kw)
            if self.enabled:   # the locations the launch computed on chip, recomputed outside the timed interval to count V
                from simpb_amd.plugin import ops
                loc = ops.dfa_locations(anchor, learn, fix_scale, proj, wh)
                lvls = ss.shape[1]
                self.daf_calls.append((loc, (loc.shape[0], loc.shape[1], loc.shape[2], loc.shape[3], lvls, groups), tuple(out.shape),
                                       dict(feat_bytes=feat.element_size(), fused=True, logits=feat_logits.numel() + cam_logits.numel(),
                                            small=anchor.numel() + learn.numel())))
            return out

        def msda_lin(tokens, ss, lsi, raw, ref, qcam, m_live=None):
            out = self.lin_orig(tokens, ss, lsi, raw, ref, qcam, m_live)
            if self.enabled:
                self.msda_calls.append(((raw.shape[0], raw.shape[1], 8, 4, 4, 2), 256, qcam,
                                        dict(tok_bytes=tokens.element_size(), row=out.shape[-1], raw=raw.shape[-1])))
            return out

        from simpb_amd.plugin import ops as _ops
        self.ops, self.lin_orig = _ops, _ops.msda_linear
        _ops.msda_linear = msda_lin
        self.blocks.DAF = daf
        self.blocks.dfa_fused = fused
        self.group_attn.ms_deform_attn_grouped = msda
        return self

    def __exit__(self, *exc):
        self.blocks.DAF, self.group_attn.ms_deform_attn_grouped = self.daf_orig, self.msda_orig
        self.blocks.dfa_fused = self.fused_orig
        self.ops.msda_linear = self.lin_orig
        self.lib.simpb_timing_enable(0)

    def start(self):
        self.lib.simpb_timing_reset()
        self.enabled = True

    def _durations(self, kid):
        import ctypes
        buf = (ctypes.c_float * self.capacity)()
        n = self.lib.simpb_timing_read(kid, buf, self.capacity)
        return [buf[i] * 1e-3 for i in range(max(n, 0))]

    def summary(self):
        self.enabled = False
        out = {}
        d = self._durations(self.DAF)
        if d and len(d) == len(self.daf_calls):
            nbytes = valid = 0.0
            kernel, s_f = "daf_fwd_rows", 4
            for call in self.daf_calls:
                loc, wshape, oshape = call[:3]
                v = int(((loc > 0) & (loc < 1)).all(-1).sum())
                out_bytes = oshape[0] * oshape[1] * oshape[2] * 4
                if len(call) > 3:
                    # one-launch form (csrc/deform_agg_fused.hip): SURVEY.md §8(d) with s_f = the token element size it reads;
                    # sampling locations and weights never exist in memory -- what it reads instead are the logits they are
                    # made from (feat_logits + cam_logits) and the anchor / learnable-offset rows
                    info = call[3]
                    kernel, s_f = "daf_fused_rows", info["feat_bytes"]
                    nbytes += v * wshape[4] * 4 * oshape[-1] * s_f + (info["logits"] + info["small"]) * 4 + out_bytes
                else:
                    w_elems = 1
                    for k in wshape:
                        w_elems *= k
                    # V * L levels * 4 taps * C * 4 B + loc + weights + out
                    nbytes += v * wshape[4] * 4 * oshape[-1] * 4 + loc.numel() * 4 + w_elems * 4 + out_bytes
                valid += v
            n = len(d)
            out["daf"] = dict(kernel=kernel, secs=sum(d) / n, nbytes=nbytes / n, launches=n, valid_triples=valid / n,
                              feature_bytes_per_element=s_f)
        d = self._durations(self.MSDA)
        if d and len(d) == len(self.msda_calls):
            nbytes = survey = 0.0
            kernel, s_v = "msda_grouped_fwd", 4
            for call in self.msda_calls:
                (bs, nq, heads, lvls, pts, _), ch, qcam = call[:3]
                # capacity slots outside every camera group (query_cam < 0) are skipped by the kernel and not counted
                nq = int((qcam >= 0).sum())
                survey += bs * nq * (heads * lvls * pts * 4 * 32 * 4 + heads * lvls * pts * 3 * 4 + 256 * 4)   # N2 x (65 536 + 1 536 + 1 024) B
                if len(call) > 3:
                    # sampling of the RAW tokens (csrc/msda_lin.hip: value_proj moved behind the sampling): per query
                    # heads*lvls*pts samples * 4 taps * 256 channels * token bytes + offsets|logits row + the 8 x 256 + tail row written
                    info = call[3]
                    kernel, s_v = "msda_linear_fwd", info["tok_bytes"]
                    nbytes += bs * nq * (heads * lvls * pts * 4 * ch * s_v + info["raw"] * 4 + 8 + info["row"] * 4)
                else:
                    # per query: heads*lvls*pts samples * 4 taps * ch * 4 B + loc + attn + out
                    nbytes += bs * nq * (heads * lvls * pts * 4 * ch * 4 + heads * lvls * pts * 3 * 4 + heads * ch * 4)
            n = len(d)
            out["msda"] = dict(kernel=kernel, secs=sum(d) / n, nbytes=nbytes / n, launches=n, feature_bytes_per_element=s_v,
                               survey_nbytes=survey / n)
        return out


def build_model(args, device):
    from simpb_amd import configs, plugin, synth
    cfg = configs.simpb_plus(depth=args.depth, input_shape=tuple(args.image_wh), anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    if args.residual_damp != 1.0:   # synthetic-weight conditioning (see --residual-damp): the last BatchNorm of every bottleneck
        with torch.no_grad():
            for m in model.img_backbone.modules():
                if hasattr(m, "bn3") and hasattr(m, "conv3"):
                    m.bn3.weight.mul_(args.residual_damp)
                    m.bn3.bias.mul_(args.residual_damp)
    model.to(device)
    model.fuse_conv_bn()  # the reference's own --fuse-conv-bn inference option (tools/benchmark.py:76-78)
    model.half_backbone()  # fp16 backbone+FPN, fp32 head: the reference's own precision split (config :26)
    with torch.no_grad():
        img = synth.images(args.bs, 0, tuple(args.image_wh)).to(device)
        std = float(model.extract_feat(img)[0].float().std())
        if args.token_std:
            for m in model.img_neck.fpn_convs:
                m.conv.weight.mul_(args.token_std / std)
                if m.conv.bias is not None:
                    m.conv.bias.mul_(args.token_std / std)
            std = float(model.extract_feat(img)[0].float().std())
    model.simpb_token_std = std
    return model


def make_frames(args, device, n_frames):
    """Pre-stage a short ring of distinct frames in HBM; timestamps keep advancing past the ring."""
    from simpb_amd import synth
    imgs = [synth.images(args.bs, f % 4, tuple(args.image_wh)).to(device) for f in range(min(n_frames, 4))]
    return imgs


def frame_metas(args, f):
    """Host-side metadata of frame f, as the reference's test pipeline collects it (config :349-358)."""
    from simpb_amd import synth
    return synth.frame_metas(args.bs, f, tuple(args.image_wh))


def cpu_baseline(args):
    """The oracle head (+ an fp32 CPU run of the same backbone/FPN) on the host cores for a
    bounded sample of the same workload. kind = 'port': the reference itself cannot travel."""
    from oracle import simpb_ref as R
    from simpb_amd import configs, plugin, synth
    # the box's CPU share (its affinity mask; a one-GPU box is entitled to 16 workers), not the machine's core count:
    # a thread pool sized to os.cpu_count() on a shared host oversubscribes the share and runs slower
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(share, 16)
    torch.set_num_threads(cores)
    cfg = configs.simpb_plus(depth=args.depth, input_shape=tuple(args.image_wh), anchor=synth.anchors(900))
    model = plugin.build_detector(cfg["model"]).eval()
    synth.load_procedural(model)
    params = {k: v.detach() for k, v in model.head.state_dict().items()}
    head = R.OracleHead(params, model.head.operation_order)
    times = []
    with torch.no_grad():
        for f in range(args.cpu_frames):
            img = synth.images(1, f % 4, tuple(args.image_wh))
            metas = synth.frame_metas(1, f, tuple(args.image_wh))
            t0 = time.perf_counter()
            feats = model.img_neck(model.img_backbone(img.flatten(end_dim=1)))
            fm = R.feature_maps_format([x.reshape((1, 6) + x.shape[1:]) for x in feats])
            outs = head.forward(fm, metas)
            head.post_process(outs, metas)
            times.append(time.perf_counter() - t0)
    warm = times[1:] or times
    return dict(value=len(warm) / sum(warm), unit="frames/s", cores=cores, kind="port",
                ms_per_frame_min=min(warm) * 1e3, ms_per_frame_max=max(warm) * 1e3,
                sample=f"bounded sample of the same workload: {len(warm)} warm frame(s) after 1 cold (BASELINE.md's protocol is 50; a "
                       f"frame takes ~1 s here), bs=1, oracle head fp32 + PyTorch CPU ResNet{args.depth}+FPN fp32, torch threads={cores} "
                       f"(os.cpu_count()={os.cpu_count()}, affinity mask={share}: the threads are the box's CPU share, capped at 16)")


def head_dtype_note():
    """What the decoder computes in, as the route table has it for this run."""
    from simpb_amd.plugin import routes
    gemm = ("grouped GEMMs on the FP16 matrix cores with split operands (x = xh + xl / 2^11, all four partial products, fp32 accumulators: "
            "error vs float64 0.45-0.6 x the exact-fp32 matrix kernel's, profiles/r04_gemm_split_error.txt)"
            if routes.R.gemm_split_fp16 else "grouped GEMMs on the exact-fp32 matrix instruction")
    att = ("attention on the FP16 matrix cores with split operands (three partial products per product, fp32 softmax; same bound vs "
           "float64 as the exact kernel, tests/test_gpu_ops.py)" if routes.R.attention_split_fp16 else "attention on the exact-fp32 matrix instruction")
    return ("f32 storage, f32 accumulation and fp32-grade products throughout (no bf16 / fp16-rounded operand anywhere in the head): "
            f"{gemm}; {att}; MLP chains and samplers in plain fp32; the camera tokens are the fp16 backbone's output, sampled as they "
            "are and accumulated in f32")


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, as the child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` (one process per GPU),
    BEFORE this process has touched the GPU, and exit with its status. The child's rank 0 prints the JSON line."""
    import socket
    import subprocess
    # device_count() may or may not create a HIP context on ROCm (it can fall back to hipGetDeviceCount): harmless here
    # because the ranks are started as a CHILD process (subprocess.run), never by replacing this one (no exec)
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("SIMPB_BENCH_DEVICE") is None:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s); refusing to report a {args.gpus}-GPU "
                         "number from fewer (set SIMPB_BENCH_DEVICE only for the one-GPU rehearsal)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus is None:   # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's world is the GPU count
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # never label a run with a GPU count it did not use
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if os.environ.get("SIMPB_BENCH_DEVICE") is not None:  # rehearsal of N > 1 on a one-GPU box
        local_rank = int(os.environ["SIMPB_BENCH_DEVICE"])
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # host-side torch ops in the frame loop are tiny; keep the intra-op pool from oversubscribing
    # the box's CPU share (a pool sized to the machine's core count stalls frames for tens of ms)
    torch.set_num_threads(max(1, min(4, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)))

    if args.lib:
        import simpb_amd._lib as _l
        _l.LIB = os.path.abspath(args.lib)
    from simpb_amd.dist import DetectionGather
    from simpb_amd.runner import FrameRunner, PipelinedRunner, SplitPipelinedRunner
    torch.backends.cudnn.benchmark = not args.no_conv_search
    # the secondary PCIe-inclusive leg (one GPU, one runner: its frames sit between the timed region and the roofline leg)
    h2d_frames = 4 + args.h2d_steps if (not args.h2d and args.h2d_steps > 0 and world == 1 and args.streams == 1
                                        and not args.no_pipeline and not args.eager) else 0
    total = args.prime + args.warmup + args.steps + h2d_frames + args.meter_frames
    imgs = make_frames(args, device, total)
    if args.h2d:  # the reference's protocol times model(**data) with the scatter to the device inside (tools/benchmark.py:90-100)
        imgs = [x.cpu().pin_memory() for x in imgs]

    def frame_of(f):
        return imgs[f % len(imgs)]
    metas = [frame_metas(args, f) for f in range(total)]  # what a dataloader would hand over
    pipelined = not args.no_pipeline and not args.eager
    if args.streams > 1 and not pipelined:
        raise SystemExit("--streams needs the pipelined runner")
    # taking the single-frame decoder layer off the temporal chain pays off when ONE stream's chain of dependent launches
    # bounds the frame (348 vs 335 frames/s); with several streams or a batch the chip is busy anyway and the extra
    # launches cost (8 streams: 368 vs 394 per GPU)
    split = pipelined and not args.no_split and args.streams == 1 and args.bs == 1
    if args.capacity_by_rank:
        args.capacity = args.capacity_by_rank[min(rank, len(args.capacity_by_rank) - 1)]
    runners = []
    for _ in range(args.streams):  # one model replica + runner per independent stream
        model = build_model(args, device)
        runners.append(((SplitPipelinedRunner if split else PipelinedRunner) if pipelined else FrameRunner)(
            model, args.bs, (args.image_wh[1], args.image_wh[0]), capacity=args.capacity, device=device,
            use_graph=not args.eager, independent_streams=not args.reference_batch))
    runner = runners[0]
    # N > 1: the fixed-shape device record of every stream to every rank, one all-gather per frame on a side stream
    gather = None
    if dist is not None:
        # 2D rows: only the slots of the kept 3D boxes travel (compacted on the device): num_output x num_cams bounds them on
        # every rank whatever its slot capacity is or grows to, so the exchange shape never has to be re-agreed
        gather = DetectionGather(args.streams * args.bs, runner.head.decoder.num_output, device,
                                 rows2d=runner.head.decoder.num_output * runner.head.num_cams, compact2d=True)

    def step(f, force_eager=False):
        if len(runners) == 1:
            results = runner.step(frame_of(f), metas[f], force_eager=force_eager)
        else:  # launch every stream's frame, then collect: the streams' work runs side by side
            for i, r in enumerate(runners):
                r.launch(frame_of(f + i), metas[f], force_eager=force_eager)
            results = None
            for r in runners:
                out = r.collect()
                if out is not None:
                    results = (results or []) + out
        if results is None:  # pipelined runner, very first call: nothing decoded yet
            return None
        if gather is not None:
            gather.submit([r.last_rec3d for r in runners], [r.s_rec for r in runners if hasattr(r, "s_rec")],
                          records2d=[r.last_rec2d for r in runners])
            for r in runners:
                r.rec_consumed = gather.done
        return results

    first = args.prime + args.warmup
    for f in range(first):
        step(f)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(first, first + args.steps):
        results = step(f)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    # secondary figure: the reference's own timing protocol (tools/benchmark.py:90-100 times model(**data) with the scatter
    # of the frame to the device inside). Never `value`: the same runner, the same streams, frames now starting in pinned
    # host memory and crossing PCIe inside the step (async copy on the backbone stream, beside the previous frame's decoder)
    h2d_leg = None
    base = first + args.steps
    if rank == 0 and world == 1 and h2d_frames and pipelined and len(runners) == 1:
        pinned = [x.cpu().pin_memory() for x in imgs]   # the same ring of frames, now in pinned host memory
        for f in range(base, base + 4):
            runner.step(pinned[f % len(pinned)], metas[f])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in range(base + 4, base + h2d_frames):
            runner.step(pinned[f % len(pinned)], metas[f])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        h2d_leg = dict(value=args.streams * args.bs * args.h2d_steps / dt, unit="frames/s", ms_per_step=dt / args.h2d_steps * 1e3,
                       steps=args.h2d_steps,
                       note="frames start in pinned host memory and cross PCIe inside the timed step (13 MB per 6-camera frame): the "
                            "reference's protocol, tools/benchmark.py:90-100; `value` above has its inputs resident in HBM")

    # roofline leg: the next frames of the same streams, launched one kernel at a time (no graph) with
    # a HIP event pair recorded around every sampler launch. Event nodes cannot be read back from
    # inside a replayed graph, so the timed region above stays uninstrumented; rocprofv3 of this same
    # command sees both and its per-kernel average is what profiles/ holds.
    ksum = {}
    if rank == 0 and args.meter_frames > 0:
        with KernelMeter(args.meter_frames) as kt:
            kt.start()
            for f in range(total - args.meter_frames, total):
                runner.step(frame_of(f), metas[f], force_eager=True)  # rank-local: no collective here
            torch.cuda.synchronize()
            ksum = kt.summary()

    t = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank != 0:   # (rank 0 prints the line; the others leave what their runners did on stderr: rehearsals look for overflows there)
        print(f"[rank {rank}] frame_runner {dict(runner.stats, capacity_2d=runner.capacity)}", file=sys.stderr, flush=True)
    if rank == 0:
        frames = world * args.streams * args.bs * args.steps
        head0 = runner.head
        n2 = [int(x) for x in head0.layers[0].last.count.sum(dim=1).tolist()] if head0.layers[0].last else None
        mode = dict(runner.stats, hipgraph=not args.eager, capacity_2d=args.capacity,
                    pipelined_backbone=pipelined, single_frame_layer_beside_previous_frame=split)
        def pmc_traffic(kernel):
            """HBM-side bytes per launch of `kernel` from the newest committed PMC passes (profiles/r*_sampler_traffic.json;
            rocprofv3 --pmc cannot run inside this process: tools/profile_round.py takes them with this same command),
            with the file they came from; (None, None) when no profile of this kernel is committed."""
            import glob
            if not (args.bs == 1 and args.depth == 50 and tuple(args.image_wh) == (704, 256)):
                return None, None, None   # the committed passes were taken on the default workload (one R50 704x256 frame per launch)
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sampler_traffic.json")), reverse=True):
                try:
                    entry = json.load(open(path))["kernels"][kernel]
                    return entry["traffic_bytes_per_launch"], os.path.relpath(path, ROOT), (entry.get("rocprof") or {}).get("avg_us")
                except (KeyError, ValueError):
                    continue
            return None, None, None

        tokens = 6 * sum((args.image_wh[1] // s) * (args.image_wh[0] // s) for s in (4, 8, 16, 32))

        def roofline(k, what):
            feat_mb = args.bs * tokens * 256 * k.get("feature_bytes_per_element", 4) / 1e6   # (a launch reads the maps of all its streams)
            in_cache = feat_mb * 1e6 < 256 * 2 ** 20
            ach = k["nbytes"] / k["secs"] / 1e9
            traffic, src, prof_us = pmc_traffic(k["kernel"])
            r = dict(kernel=k["kernel"], bound="hbm", achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s",
                     frac=ach / HBM_PEAK_GBPS, traffic=traffic, avg_us=k["secs"] * 1e6, algorithmic_MB=k["nbytes"] / 1e6,
                     launches=k["launches"],
                     regime="infinity-cache" if in_cache else "hbm",
                     hbm_side_GBps=(traffic / k["secs"] / 1e9) if traffic else None,
                     hbm_side_frac=(traffic / k["secs"] / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                     traffic_source=src,
                     # the same algorithmic bytes over the kernel's average duration under rocprofv3 --kernel-trace --stats of
                     # this command (committed with the PMC passes; no launch latency in it, unlike the event interval above)
                     rocprof_avg_us=prof_us, frac_at_rocprof_time=(k["nbytes"] / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if prof_us else None,
                     note=f"{what}; achieved = algorithmic bytes (SURVEY.md 8d) / launch time measured by HIP events on the "
                          f"launch stream in {k['launches']} instrumented launches right after the timed region; the feature "
                          f"set it reads is {feat_mb:.1f} MB and " + ("fits the 256 MiB Infinity Cache, so `achieved` is an on-die rate and can "
                          "exceed the HBM peak: hbm_side_* = PMC bytes beyond L2 (2*FETCH_SIZE + WRITE_SIZE) / the same time"
                          if in_cache else "does not fit the 256 MiB Infinity Cache; `achieved` counts every tap row as its own read "
                          "(SURVEY.md 8d), and the four taps of a sample and neighbouring samples share L2 lines, so it can still exceed the HBM peak"))
            if "valid_triples" in k:
                r["valid_triples"] = k["valid_triples"]
            if "survey_nbytes" in k:
                # SURVEY.md 8(d)'s own byte count for the operator (the reference's head-slice sampler: 128-B fp32 slices of the
                # PROJECTED map) over this kernel's time: the kernel gathers 4 x those bytes on purpose (whole 512-B f16 rows of
                # the RAW tokens per head, which is what deletes the 11.8 GFLOP value_proj per layer), so `frac` above is a
                # cache rate of the bytes it really touches and THIS is the fraction the survey's figure is priced at
                r["survey_algorithmic_MB"] = k["survey_nbytes"] / 1e6
                r["frac_survey_bytes"] = k["survey_nbytes"] / k["secs"] / 1e9 / HBM_PEAK_GBPS
                r["frac_survey_bytes_at_rocprof_time"] = (k["survey_nbytes"] / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if prof_us else None
            return r

        roof = roofline(ksum["daf"], f"3D deformable aggregation ({ksum['daf']['kernel']}, {ksum['daf'].get('feature_bytes_per_element', 4)} B per token element)") if "daf" in ksum else None
        roof2 = roofline(ksum["msda"], "camera-grouped MSDeformAttn sampling of the raw camera tokens (value_proj applied behind the sampling)"
                         if ksum["msda"]["kernel"] == "msda_linear_fwd" else "camera-grouped MSDeformAttn sampling over the value_proj output") if "msda" in ksum else None
        line = {
            "metric": "frames/sec (6-cam sample) + MSDeformAttn HBM GB/s, R50 704x256 @1/2/4/8 GPU",
            "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"simpb_nus_r{args.depth}_img_{args.image_wh[0]}x{args.image_wh[1]}: 6-cam frames, "
                                   f"ResNet{args.depth}+FPN (own HIP convolutions, fp16) + HIP decoder, PyTorch-ROCm as the host / memory / stream layer, "
                                   f"{args.streams} stream(s) x bs={args.bs} per GPU, temporal",
                       "streams_per_gpu": args.bs * args.streams, "batch_semantics": ("one stream" if args.bs == 1 else "reference batch (groups padded to the max over the batch)" if args.reference_batch else f"{args.bs} independent streams per launch, each decoded as a batch of one (SURVEY.md 8e)"), "backbone_dtype": "f16 (backbone+FPN only, the reference's own fp16 split: config :26, simpb.py:63)", "head_dtype": head_dtype_note(), "parallelism": f"stream-sharded x{world}, RCCL all-gather of detections"
                       if world > 1 else "single GPU", "inputs": "pinned host frames, H2D inside the timed step" if args.h2d else "resident in HBM",
                       "fp16_matrix_step": "2 x v_mfma_f32_32x32x8f16 (csrc/mfma_f16.h)", "num_query2d_last_frame": n2, "frame_runner": mode,
                       "token_std": getattr(runner.model, "simpb_token_std", None)},
            "roofline": roof, "roofline_msda": roof2, "reference_protocol_h2d": h2d_leg,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    _routes = [a for i, a in enumerate(sys.argv) if i and sys.argv[i - 1] == "--route"]
    if _routes:   # (measurement only: the switches are read-only outside this context manager)
        from simpb_amd.plugin import routes
        with routes.override(**{k: bool(int(v)) for k, v in (r.split("=") for r in _routes)}):
            main()
    else:
        main()
