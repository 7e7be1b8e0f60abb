"""Frame runner: drives SimPB frame by frame for a fixed set of streams and replays the warm
(temporal) frame as ONE hipGraph.

Why: in eager mode a frame is ~1 700 kernel launches and the host, not the GPU, sets the pace. With
static shapes (a fixed-capacity 2D query set whose group table stays on the device, a temporal bank
in persistent buffers, a fixed-shape detection record) a frame has no host round trip inside it, so
it is captured once and replayed: per frame the host copies a few hundred bytes of metadata into
device buffers, launches the graph, and reads back two small records.

Frame 0 of a stream set has no history (a different dataflow) and runs eagerly; the first warm frame
runs eagerly too (lazy initialisation outside the capture), the second is captured. If a frame's 2D
query set ever exceeds the capacity, the overflow flags (read back with the records) trigger an eager
rerun of that frame with a larger capacity: the frame-end commit of the instance bank holds back when
a flag is set (csrc/bank.hip `hold`), so the rerun starts from the state the frame found, the graphs
are re-captured at the new capacity, and results never silently degrade. PipelinedRunner enqueues every
decoder one step early (before the previous frame's flags have reached the host) and therefore chains the
hold on the device: see its docstring.
"""
import numpy as np
import torch

from .plugin.detection3d import SparseBox3DDecoder


# Captures are taken in thread-local mode: with a process group initialised (one process per GPU, dist.py) the collective
# backend's watchdog thread polls events in the background, and in the default global mode a call of that kind from ANY thread
# invalidates a capture in progress. Everything a capture itself does happens on the capturing thread.
CAPTURE_MODE = "thread_local"


class FrameRunner:
    def __init__(self, model, batch_size, image_hw, capacity=1536, device=None, use_graph=True, independent_streams=False):
        """independent_streams: the batch is a set of independent camera streams, each decoded exactly as a batch of one
        would be (`capacity` 2D slots per stream; SimPBHead.independent_streams) -- the throughput form of BASELINE config
        #3. False: the reference's batch semantics (camera groups padded to the max over the batch)."""
        self.model = model
        self.head = model.head
        self.bs = batch_size
        self.independent = bool(independent_streams) and batch_size > 1
        self.head.independent_streams = self.independent
        self.capacity = int(capacity)
        self.use_graph = use_graph
        self.device = device if device is not None else next(model.parameters()).device
        h, w = image_hw
        dev = self.device
        cams = self.head.num_cams
        self.img = torch.zeros(batch_size, cams, 3, h, w, device=dev)
        self.proj = torch.zeros(batch_size, cams, 4, 4, device=dev)
        self.wh = torch.tensor([float(w), float(h)], device=dev).view(1, 1, 2).repeat(batch_size, cams, 1)
        self.wh_host = (int(w), int(h))
        self.t_buf = torch.zeros(batch_size, 4, 4, device=dev)
        self.dt_buf = torch.zeros(batch_size, device=dev)
        self.pin_t = torch.zeros(batch_size, 4, 4).pin_memory()
        self.pin_dt = torch.zeros(batch_size).pin_memory()
        self.pin_proj = torch.zeros(batch_size, cams, 4, 4).pin_memory()
        self.head.instance_bank.enable_static(batch_size, dev)
        self.head.static_capacity = self.capacity
        self.prev_metas = None
        self.graph = None
        self.outputs = None
        self.warm_frames = 0
        self.host3d = self.host2d = self.host_flag = None
        self.stats = dict(eager=0, replay=0, overflow=0)
        self.last_rec3d = None      # device record f32 [bs, num_output, 15] of the frame last returned (dist.DetectionGather)
        self.last_rec2d = None      # ... and its 2D record f32 [bs, capacity, 8]
        # event after which those records may be overwritten (set by their consumer, if any). The records of a replayed
        # frame live in graph memory: EVERY graph of this runner that may write that memory waits for it before its next
        # replay (the decoder graphs and, in SplitPipelinedRunner, part A, which shares the pool of part B)
        self.rec_consumed = None

    # ------------------------------------------------------------------ per-frame host work
    def _stage(self, img, metas):
        """Copy this frame's inputs into the static device buffers (a few small async copies)."""
        self.img.copy_(img, non_blocking=True)
        self.pin_proj.copy_(metas["projection_mat"] if not metas["projection_mat"].is_cuda else metas["projection_mat"].cpu())
        self.proj.copy_(self.pin_proj, non_blocking=True)
        if self.prev_metas is not None:
            for i, m in enumerate(metas["img_metas"]):
                t = m["T_global_inv"] @ self.prev_metas["img_metas"][i]["T_global"]  # instance_bank.py:90-97
                self.pin_t[i] = torch.from_numpy(np.asarray(t, np.float32))
                self.pin_dt[i] = float(m["timestamp"] - self.prev_metas["img_metas"][i]["timestamp"])
            self.t_buf.copy_(self.pin_t, non_blocking=True)
            self.dt_buf.copy_(self.pin_dt, non_blocking=True)

    def _device_metas(self, metas):
        out = dict(projection_mat=self.proj, image_wh=self.wh, image_wh_host=self.wh_host, img_metas=metas["img_metas"])
        if self.prev_metas is not None:
            out["bank_inputs"] = (self.t_buf, self.dt_buf)
        return out

    def _frame(self, dmetas, aug_config):
        """The device part of one frame; every tensor it returns has a fixed shape."""
        feature_maps = self.model.extract_feat(self.img)
        outs = self.head(feature_maps, dmetas)
        alloc = outs["alloc_list"][-1]
        rec3d, rec2d = self.head.decoder.decode_static_device(
            outs["classification"], outs["prediction"], outs["instance_id"], outs["quality"],
            outs["classification2d"], outs["prediction2d"], alloc, aug_config)
        return rec3d, rec2d, outs["overflow"]

    # ------------------------------------------------------------------ capacity overflow
    def _drop_graphs(self):
        self.graph = self.outputs = None

    def _grow(self):
        """A frame's 2D query set did not fit: next capacity (x1.5, multiple of 128, at most anchors x cameras, which
        no set can exceed), graphs dropped. The bank still holds the state the frame found (see module docstring)."""
        bank = self.head.instance_bank
        limit = -(-bank.num_anchor * self.head.num_cams // 128) * 128
        if not bank._fusable(bank._static["cached_anchor"]) or self.capacity >= limit:
            raise RuntimeError(
                f"2D query set exceeded the static capacity {self.capacity} and the frame cannot be re-run "
                "(the instance bank is not on its fused path, so this frame's state is already committed)")
        self.capacity = min(limit, -(-max(self.capacity + 128, self.capacity * 3 // 2) // 128) * 128)
        self.head.static_capacity = self.capacity
        self.stats["overflow"] += 1
        self._drop_graphs()

    def _read_back(self, rec3d, rec2d, flags):
        if self.host3d is None or self.host2d.shape != rec2d.shape:
            self.host3d = torch.empty(rec3d.shape, dtype=rec3d.dtype).pin_memory()
            self.host2d = torch.empty(rec2d.shape, dtype=rec2d.dtype).pin_memory()
            self.host_flag = torch.empty(flags.shape, dtype=flags.dtype).pin_memory()
        self.host3d.copy_(rec3d, non_blocking=True)
        self.host2d.copy_(rec2d, non_blocking=True)
        self.host_flag.copy_(flags, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.host3d, self.host2d, bool(self.host_flag.any())

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def step(self, img, metas, force_eager=False):
        """One frame for all streams: img f32 [bs, cams, 3, H, W] (device), metas as the reference's
        test pipeline collects them (projection_mat, timestamp, img_metas with T_global/T_global_inv/
        aug_config). Returns the reference's list of {'img_bbox': {...}} (simpb_head.py:1089-1123)."""
        aug = metas["img_metas"][0]["aug_config"]
        self._stage(img, metas)
        dmetas = self._device_metas(metas)
        warm = self.prev_metas is not None
        if warm and self.use_graph and not force_eager and self.graph is None and self.warm_frames >= 1:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                self.outputs = self._frame(dmetas, aug)
        if warm and self.graph is not None and not force_eager:
            if self.rec_consumed is not None:
                torch.cuda.current_stream().wait_event(self.rec_consumed)
            self.graph.replay()
            rec = self.outputs
            self.stats["replay"] += 1
        else:
            rec = self._frame(dmetas, aug)
            self.stats["eager"] += 1
        if warm:
            self.warm_frames += 1
        rec3d, rec2d, overflow = self._read_back(*rec)
        while overflow:  # re-run the frame on the untouched bank state with a larger 2D slot array
            self._grow()
            if not warm:
                self.head.instance_bank.reset()  # a cold frame starts from an empty bank again
            rec = self._frame(dmetas, aug)
            self.stats["eager"] += 1
            rec3d, rec2d, overflow = self._read_back(*rec)
        self.last_rec3d, self.last_rec2d = rec[0], rec[1]
        self.prev_metas = dict(img_metas=metas["img_metas"])
        self.head.instance_bank.metas = self.prev_metas
        results = SparseBox3DDecoder.decode_static_host(rec3d.numpy(), rec2d.numpy(), self.head.num_cams, self.independent)
        return [{"img_bbox": r} for r in results]


class PipelinedRunner(FrameRunner):
    """FrameRunner with the backbone of frame t+1 overlapped with the decoder of frame t.

    The decoder of frame t needs the bank the decoder of frame t-1 wrote, so decoders cannot overlap
    each other; but backbone+FPN of the next frame depends on nothing but its images. Many decoder
    kernels are small and leave most of the 256 CUs idle, while the convolutions fill the chip, so the
    two run side by side on separate HIP streams: step(t) launches backbone(t) and returns the detections of
    frame t-1 (one frame of latency for throughput; flush() returns the last frame). Two feature buffers
    alternate; each (backbone, decoder) x (buffer) pair is its own hipGraph once warm, so the steady state is two
    graph launches per step.

    The decoder of frame t is ENQUEUED in step(t) as well, behind backbone(t) (an event) and behind decoder(t-1)
    (stream order): when step(t) returns with the detections of t-1, the decoder stream already holds its next
    job, and the ~0.25 ms of host work between two steps (record finish, metadata staging, graph launches) no
    longer idles it. That decoder runs before the host has seen decoder(t-1)'s overflow flags, so the bank commit
    is chained on the device: a frame's commit also holds back when the flags of the frame enqueued before it are
    set (`overflow_chain`, plugin/head.py), and collect() then re-runs both, in order, on the state frame t-1 found."""

    def __init__(self, model, batch_size, image_hw, capacity=1536, device=None, use_graph=True, independent_streams=False):
        # two streams side by side from here on: a convolution that misses the in-tree kernels' shape rules must not slip to a
        # vendor kernel silently (plugin/detector.py: STRICT_NO_VENDOR)
        from .plugin import detector
        detector.STRICT_NO_VENDOR = True
        super().__init__(model, batch_size, image_hw, capacity, device, use_graph, independent_streams)
        dev = self.device
        # the decoder of frame t is the critical path (a chain of ~170 dependent small launches); the
        # backbone of frame t+1 only has to be done by the time that chain ends: decoder stream first
        prio = getattr(self, "STREAM_PRIORITIES", (0, -1))
        self.s_bb = torch.cuda.Stream(device=dev, priority=prio[0])
        self.s_head = torch.cuda.Stream(device=dev, priority=prio[1])
        self.s_rec = self.s_head            # the stream the detection records are written on (for their consumers)
        self.imgs = [self.img, torch.zeros_like(self.img)]
        self.fm = [None, None]              # feature maps of the frame last produced into each slot
        self.bb_graph = [None, None]
        self.bb_out = [None, None]
        self.bb_runs = [0, 0]
        self.head_graph = [None, None]
        self.head_out = [None, None]
        self.head_runs = [0, 0]
        self.count = 0
        n_alloc = sum(op == "allocation" for op in self.head.operation_order)
        self.flags = torch.zeros(2, n_alloc, dtype=torch.int32, device=dev)   # overflow flags, one row per feature slot
        self.queue = []                     # decoders in flight, oldest first: dict(slot, metas, prev, warm, rec)
        self.last_metas = None              # metas of the frame whose decoder was enqueued last
        self.bb_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.staged = None                  # event: the pinned staging buffers have been copied to the device
        self.host = [None, None]            # pinned read-back buffers per slot: (rec3d, rec2d, flags)

    def _run_backbone(self, slot, force_eager):
        """Enqueue backbone+FPN of the image in slot `slot` on s_bb."""
        with torch.cuda.stream(self.s_bb):
            if self.use_graph and not force_eager and self.bb_graph[slot] is None and self.bb_runs[slot] >= 1:
                self.s_bb.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_bb, capture_error_mode=CAPTURE_MODE):
                    self.bb_out[slot] = self._features(slot)
                self.bb_graph[slot] = g
                self.head_graph[slot] = None  # a decoder graph bound to the old buffer is stale
                self.head_runs[slot] = 0
            if self.bb_graph[slot] is not None and not force_eager:
                self.bb_graph[slot].replay()
                self.fm[slot] = self.bb_out[slot]
            else:
                self.fm[slot] = self._features(slot)
            self.bb_runs[slot] += 1

    def _features(self, slot):
        """Backbone + FPN + token format, plus everything of the decoder that depends on the features
        alone: the value projections of the three 2D cross-attention layers."""
        fm = list(self.model.extract_feat(self.imgs[slot]))
        if hasattr(self.head, "precompute_values") and len(fm) == 3:
            fm.append(self.head.precompute_values(fm))
        return fm

    def _decode(self, fm, dmetas, aug):
        outs = self.head(fm, dmetas)
        alloc = outs["alloc_list"][-1]
        rec3d, rec2d = self.head.decoder.decode_static_device(
            outs["classification"], outs["prediction"], outs["instance_id"], outs["quality"],
            outs["classification2d"], outs["prediction2d"], alloc, aug)
        return rec3d, rec2d, outs["overflow"]

    def _drop_graphs(self):
        self.head_graph, self.head_out, self.head_runs = [None, None], [None, None], [0, 0]

    def _run_head(self, slot, dmetas, aug, warm, force_eager):
        """Enqueue the decoder of the frame whose features sit in slot `slot` on s_head."""
        with torch.cuda.stream(self.s_head):
            graph_ok = (self.use_graph and not force_eager and warm and self.bb_graph[slot] is not None
                        and self.fm[slot] is self.bb_out[slot])
            if graph_ok and self.head_graph[slot] is None and self.head_runs[slot] >= 1:
                self.s_head.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.s_head, capture_error_mode=CAPTURE_MODE):
                    self.head_out[slot] = self._decode(self.fm[slot], dmetas, aug)
                self.head_graph[slot] = g
            if graph_ok and self.head_graph[slot] is not None:
                if self.rec_consumed is not None:  # the record buffer of this graph may still be read by its consumer
                    self.s_head.wait_event(self.rec_consumed)
                self.head_graph[slot].replay()
                rec = self.head_out[slot]
                self.stats["replay"] += 1
            else:
                rec = self._decode(self.fm[slot], dmetas, aug)
                self.stats["eager"] += 1
                if graph_ok:
                    self.head_runs[slot] += 1
            return rec

    def _enqueue_readback(self, slot, rec):
        with torch.cuda.stream(self.s_head):
            h = self.host[slot]
            if h is None or h[1].shape != rec[1].shape:
                h = tuple(torch.empty(r.shape, dtype=r.dtype).pin_memory() for r in rec)
                self.host[slot] = h
            for dst, src in zip(h, rec):
                dst.copy_(src, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.s_head)
        return done

    def _stage_head_inputs(self, metas, prev):
        """Per-frame decoder inputs (projection matrices, ego-motion, time step) of a frame; prev = the metas of the
        frame before it (None for a cold frame)."""
        if self.staged is not None:
            self.staged.synchronize()   # the previous frame's copies out of the pinned buffers (long done in practice)
        with torch.cuda.stream(self.s_head):
            self.pin_proj.copy_(metas["projection_mat"] if not metas["projection_mat"].is_cuda else metas["projection_mat"].cpu())
            self.proj.copy_(self.pin_proj, non_blocking=True)
            if prev is not None:
                for i, m in enumerate(metas["img_metas"]):
                    t = m["T_global_inv"] @ prev["img_metas"][i]["T_global"]
                    self.pin_t[i] = torch.from_numpy(np.asarray(t, np.float32))
                    self.pin_dt[i] = float(m["timestamp"] - prev["img_metas"][i]["timestamp"])
                self.t_buf.copy_(self.pin_t, non_blocking=True)
                self.dt_buf.copy_(self.pin_dt, non_blocking=True)
            self.staged = torch.cuda.Event()
            self.staged.record(self.s_head)

    def _head_metas(self, metas, slot, warm):
        out = dict(projection_mat=self.proj, image_wh=self.wh, image_wh_host=self.wh_host, img_metas=metas["img_metas"],
                   overflow_chain=(self.flags, slot))
        if warm:
            out["bank_inputs"] = (self.t_buf, self.dt_buf)
        return out

    def _enqueue_decoder(self, slot, metas, prev, force_eager):
        """Stage the inputs of the frame whose features sit in `slot` and enqueue its decoder + read-back on s_head."""
        warm = prev is not None
        self.prev_metas = prev   # (what the base class's helpers look at)
        self._stage_head_inputs(metas, prev)
        rec = self._run_head(slot, self._head_metas(metas, slot, warm), metas["img_metas"][0]["aug_config"], warm, force_eager)
        done = self._enqueue_readback(slot, rec)
        return dict(slot=slot, metas=metas, prev=prev, warm=warm, rec=rec, done=done)

    @torch.no_grad()
    def launch(self, img, metas, force_eager=False):
        """Enqueue backbone(t) and decoder(t) without waiting for either (several runners -- several independent
        camera streams on one GPU -- can be launched back to back and collected after)."""
        slot = self.count % 2
        cur = torch.cuda.current_stream()
        self.s_bb.wait_stream(cur)
        self.s_head.wait_stream(cur)
        with torch.cuda.stream(self.s_bb):
            self.imgs[slot].copy_(img, non_blocking=True)
        self._run_backbone(slot, force_eager)
        self.bb_done[slot].record(self.s_bb)
        self.s_head.wait_event(self.bb_done[slot])
        prev = dict(img_metas=self.last_metas["img_metas"]) if self.last_metas is not None else None
        self.queue.append(self._enqueue_decoder(slot, metas, prev, force_eager))
        self.last_metas = metas
        self.count += 1

    def _finish(self, job):
        """Wait for a decoder in flight, re-run it (and whatever was enqueued behind it) if its 2D set overflowed,
        return its detections."""
        job["done"].synchronize()
        h = self.host[job["slot"]]
        if bool(h[2].any()):
            # overflow: this frame's bank commit held back, and so did the commit of the frame enqueued behind it
            # (overflow_chain); the features of both still sit in their slots -> re-run them in order, eagerly, with a
            # larger slot array (the graphs are re-captured at the new capacity afterwards)
            behind = list(self.queue)
            self.queue.clear()
            self._quiesce()
            while bool(h[2].any()):
                self._grow()
                if not job["warm"]:
                    self.head.instance_bank.reset()  # a cold frame starts from an empty bank again
                self._clear_hold()   # flags left by the overflowed attempt / the speculative decoder behind it
                job = self._enqueue_decoder(job["slot"], job["metas"], job["prev"], True)
                job["done"].synchronize()
                h = self.host[job["slot"]]
            for b in behind:
                self.queue.append(self._enqueue_decoder(b["slot"], b["metas"], b["prev"], True))
        self.last_rec3d, self.last_rec2d = job["rec"][0], job["rec"][1]
        self.prev_metas = dict(img_metas=job["metas"]["img_metas"])
        results = SparseBox3DDecoder.decode_static_host(h[0].numpy(), h[1].numpy(), self.head.num_cams, self.independent)
        return [{"img_bbox": r} for r in results]

    def _quiesce(self):
        self.s_head.synchronize()

    def _clear_hold(self):
        with torch.cuda.stream(self.s_head):
            self.flags.zero_()

    def collect(self):
        """Returns the detections of frame t-1 (None the first time): waits for the decoder enqueued one step ago,
        not for what launch() just enqueued."""
        if len(self.queue) < 2:
            return None
        return self._finish(self.queue.pop(0))

    def step(self, img, metas, force_eager=False):
        """Feed frame t; returns the detections of frame t-1 (None on the very first call)."""
        self.launch(img, metas, force_eager)
        return self.collect()

    @torch.no_grad()
    def flush(self):
        """Wait for the decoder of the last fed frame and return its detections."""
        out = None
        while self.queue:
            out = self._finish(self.queue.pop(0))
        return out


def refinement_time_step(dt, max_time_interval, default_time_interval):
    """The time step the refinement heads divide velocities by (instance_bank.py:87,108-113; csrc/bank.hip bank_get_kernel):
    the gap to the previous frame where it is usable (non-zero and within max_time_interval), the default otherwise. f32
    in, f32 out, compared in f32 like the kernel does."""
    dt = np.asarray(dt, np.float32)
    ok = (dt != 0) & (np.abs(dt) <= np.float32(max_time_interval))
    return np.where(ok, dt, np.float32(default_time_interval)).astype(np.float32)


class SplitPipelinedRunner(PipelinedRunner):
    """PipelinedRunner with the single-frame decoder layer taken off the temporal chain.

    Frame t's decoder needs the bank frame t-1 committed -- but not from its first instruction: the first decoder layer
    (`num_single_frame_decoder`, simpb_head.py:690-696: allocation, the 2D block, aggregation, the first refinement) starts
    from the learned anchors alone, and the bank enters with InstanceBank.update behind it. SimPBHead.forward_split pauses
    there. This runner replays that first part ("A", ~1/6 of the decoder) on the backbone stream right behind backbone(t),
    beside the temporal part ("B") of frame t-1; B(t) then waits for A(t) (an event) and B(t-1) (stream order). The chain
    of dependent launches a frame adds to the critical path shrinks by A.

    What changes with A(t) running while B(t-1) is still in flight:
      * per-frame decoder inputs (projection matrices, ego-motion, time step) get one device buffer per feature slot;
      * the overflow hold is chained through a `sticky` word that B writes at its end (A(t+1) must not be able to disturb
        the flags B(t) looks at): SimPBHead.forward_split, `overflow_split`;
      * eager (warm-up, re-run) frames run A and B back to back on the decoder stream: only replayed graphs run A on the
        backbone stream, so no tensor of the caching allocator crosses streams.
    """

    def __init__(self, model, batch_size, image_hw, capacity=1536, device=None, use_graph=True, independent_streams=False):
        super().__init__(model, batch_size, image_hw, capacity, device, use_graph, independent_streams)
        dev = self.device
        # part A rides on the backbone stream, right behind backbone(t): as fast for one stream as a third stream of its own
        # (350 frames/s either way) and cheaper when several runners share the GPU (8 runners: 368 against 308 frames/s)
        self.s_pre = self.s_bb
        n_alloc = self.flags.shape[1]
        self.hb = torch.zeros(2, n_alloc + 1, dtype=torch.int32, device=dev)   # per slot: the frame's flags | sticky copy
        self.sticky = torch.zeros(1, dtype=torch.int32, device=dev)
        cams = self.head.num_cams
        self.proj2 = [torch.zeros(batch_size, cams, 4, 4, device=dev) for _ in range(2)]
        self.t_buf2 = [torch.zeros(batch_size, 4, 4, device=dev) for _ in range(2)]
        self.dt_buf2 = [torch.zeros(batch_size, device=dev) for _ in range(2)]
        self.ti_buf2 = [torch.zeros(batch_size, device=dev) for _ in range(2)]
        self.pin2 = [dict(proj=torch.zeros(batch_size, cams, 4, 4).pin_memory(), t=torch.zeros(batch_size, 4, 4).pin_memory(),
                          dt=torch.zeros(batch_size).pin_memory(), ti=torch.zeros(batch_size).pin_memory()) for _ in range(2)]
        self.staged2 = [None, None]
        self.pre_graph = [None, None]
        self.pre_done = [torch.cuda.Event(), torch.cuda.Event()]

    def _drop_graphs(self):
        super()._drop_graphs()
        self.pre_graph = [None, None]

    def _stage_slot(self, slot, metas, prev, stream):
        """Per-frame decoder inputs of the frame in `slot`, into that slot's own device buffers."""
        if self.staged2[slot] is not None:
            self.staged2[slot].synchronize()
        pin = self.pin2[slot]
        bank = self.head.instance_bank
        pin["proj"].copy_(metas["projection_mat"] if not metas["projection_mat"].is_cuda else metas["projection_mat"].cpu())
        pin["ti"].fill_(float(bank.default_time_interval))
        if prev is not None:
            for i, m in enumerate(metas["img_metas"]):
                t = m["T_global_inv"] @ prev["img_metas"][i]["T_global"]
                pin["t"][i] = torch.from_numpy(np.asarray(t, np.float32))
                pin["dt"][i] = float(m["timestamp"] - prev["img_metas"][i]["timestamp"])
            # the time step the refinement heads divide by (instance_bank.py:108-113, csrc/bank.hip bank_get_kernel): the
            # frame gap where it is usable, the default otherwise -- a function of the time stamps alone, in f32 like there
            pin["ti"].copy_(torch.from_numpy(refinement_time_step(pin["dt"].numpy(), bank.max_time_interval,
                                                                  bank.default_time_interval)))
        with torch.cuda.stream(stream):
            self.proj2[slot].copy_(pin["proj"], non_blocking=True)
            self.ti_buf2[slot].copy_(pin["ti"], non_blocking=True)
            if prev is not None:
                self.t_buf2[slot].copy_(pin["t"], non_blocking=True)
                self.dt_buf2[slot].copy_(pin["dt"], non_blocking=True)
            self.staged2[slot] = torch.cuda.Event()
            self.staged2[slot].record(stream)

    def _split_metas(self, metas, slot, warm):
        out = dict(projection_mat=self.proj2[slot], image_wh=self.wh, image_wh_host=self.wh_host, img_metas=metas["img_metas"],
                   time_interval=self.ti_buf2[slot], overflow_split=(self.hb[slot], self.sticky))
        if warm:
            out["bank_inputs"] = (self.t_buf2[slot], self.dt_buf2[slot])
        return out

    def _part_a(self, slot, dmetas):
        gen = self.head.forward_split(self.fm[slot], dmetas)
        next(gen)
        return gen

    def _part_b(self, gen, aug):
        try:
            gen.send(None)
        except StopIteration as done:
            outs = done.value
        else:
            raise RuntimeError("forward_split paused twice")
        alloc = outs["alloc_list"][-1]
        rec3d, rec2d = self.head.decoder.decode_static_device(
            outs["classification"], outs["prediction"], outs["instance_id"], outs["quality"],
            outs["classification2d"], outs["prediction2d"], alloc, aug)
        return rec3d, rec2d, outs["overflow"]

    def _enqueue_decoder(self, slot, metas, prev, force_eager):
        warm = prev is not None
        self.prev_metas = prev
        aug = metas["img_metas"][0]["aug_config"]
        dmetas = self._split_metas(metas, slot, warm)
        graph_ok = (self.use_graph and not force_eager and warm and self.bb_graph[slot] is not None
                    and self.fm[slot] is self.bb_out[slot])
        if graph_ok and self.head_graph[slot] is None and self.head_runs[slot] >= 1:
            # capture A and B of this slot (both on the decoder stream; A is replayed on the backbone stream afterwards)
            self.s_pre.synchronize()
            self.s_head.synchronize()
            self._stage_slot(slot, metas, prev, self.s_head)
            with torch.cuda.stream(self.s_head):
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=self.s_head, capture_error_mode=CAPTURE_MODE):
                    gen = self._part_a(slot, dmetas)
                with torch.cuda.graph(gb, stream=self.s_head, pool=ga.pool(), capture_error_mode=CAPTURE_MODE):
                    self.head_out[slot] = self._part_b(gen, aug)
                del gen
            self.pre_graph[slot], self.head_graph[slot] = ga, gb
        if graph_ok and self.head_graph[slot] is not None:
            self._stage_slot(slot, metas, prev, self.s_pre)
            with torch.cuda.stream(self.s_pre):
                self.s_pre.wait_event(self.bb_done[slot])
                if self.rec_consumed is not None:   # part A shares part B's pool: the records may sit in memory A reuses
                    self.s_pre.wait_event(self.rec_consumed)
                self.pre_graph[slot].replay()
                self.pre_done[slot].record(self.s_pre)
            with torch.cuda.stream(self.s_head):
                self.s_head.wait_event(self.pre_done[slot])
                if self.rec_consumed is not None:
                    self.s_head.wait_event(self.rec_consumed)
                self.head_graph[slot].replay()
            rec = self.head_out[slot]
            self.stats["replay"] += 1
        else:
            self.s_head.wait_stream(self.s_pre)   # (covers backbone(t) and a replayed A of the other slot)
            self._stage_slot(slot, metas, prev, self.s_head)
            with torch.cuda.stream(self.s_head):
                rec = self._part_b(self._part_a(slot, dmetas), aug)
            self.stats["eager"] += 1
            if graph_ok:
                self.head_runs[slot] += 1
        done = self._enqueue_readback(slot, rec)
        return dict(slot=slot, metas=metas, prev=prev, warm=warm, rec=rec, done=done)

    def _quiesce(self):
        self.s_pre.synchronize()
        self.s_head.synchronize()

    def _clear_hold(self):
        # an overflowed attempt leaves its flags and sticky = 1 (so does the speculative decoder behind it): every re-run
        # starts clean
        with torch.cuda.stream(self.s_head):
            self.sticky.zero_()
            self.hb.zero_()
