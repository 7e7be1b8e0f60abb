"""Multi-GPU layer of the path: camera streams are independent (frame t of a
stream reads the bank frame t-1 of the SAME stream wrote), so streams are sharded across ranks with
no collective inside a frame; the one exchange is an all-gather of a fixed-shape detection record
per frame (backend 'nccl' = RCCL over xGMI on the GPU box, 'gloo' in the CPU tests). Replaces the
reference's pickle-to-tmpdir + barrier collection (apis/test.py:122-171)."""
import numpy as np
import torch
import torch.distributed as dist

# 10 box + score + label + cls_score + the int64 instance id as two bit-cast 32-bit lanes (decoder.py:230-251;
# include/simpb_hip.h: SIMPB_RECORD3D_WIDTH). This is the record csrc/decode.hip leaves on the device.
RECORD_WIDTH = 15
ID_LANES = slice(13, 15)


def ids_to_lanes(ids):
    """int64 [...] -> f32 [..., 2] holding the same 8 bytes (low dword first). A float32 is exact only up to 2^24;
    the reference keeps track ids int64 end to end (decoder.py:247-251, instance_bank.py:169-184)."""
    return ids.contiguous().to(torch.int64).view(torch.float32).reshape(tuple(ids.shape) + (2,))


def lanes_to_ids(lanes):
    """f32 [..., 2] (torch or numpy) -> int64 [...]: the inverse of ids_to_lanes, bit for bit."""
    if isinstance(lanes, np.ndarray):
        return np.ascontiguousarray(lanes, dtype=np.float32).view(np.int64)[..., 0]
    return lanes.contiguous().view(torch.int64)[..., 0]


def shard_streams(num_streams, rank, world):
    """Contiguous block of stream ids for this rank (the reference shards whole scenes
    contiguously per rank: datasets/samplers/distributed_sampler.py:61-79)."""
    per, rem = divmod(num_streams, world)
    start = rank * per + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))


def pack_detections(results, device=None, num_output=300):
    """list of per-stream result dicts (head.post_process) -> f32 [streams, num_output, 15] in the layout of the
    device record; rows past a stream's own count (score_threshold set) are zero with id -1. Host-side route for
    callers that only hold the reference's dicts; the runners hand over the device record itself (DetectionGather)."""
    recs = []
    for r in results:
        d = r["img_bbox"] if "img_bbox" in r else r
        n = d["boxes_3d"].shape[0]
        rec = torch.zeros(num_output, RECORD_WIDTH)
        ids = torch.full((num_output,), -1, dtype=torch.int64)
        ids[:n] = d["instance_ids"].cpu().to(torch.int64)
        rec[:n, :10] = d["boxes_3d"].cpu()
        rec[:n, 10] = d["scores_3d"].cpu()
        rec[:n, 11] = d["labels_3d"].cpu().float()
        rec[:n, 12] = d["cls_scores"].cpu()
        rec[:, ID_LANES] = ids_to_lanes(ids)
        recs.append(rec)
    out = torch.stack(recs)
    return out if device is None else out.to(device, non_blocking=True)


def unpack_detections(record):
    """f32 [..., num_output, 15] -> dict of tensors (boxes_3d, scores_3d, labels_3d, cls_scores, instance_ids int64)."""
    record = record.cpu() if torch.is_tensor(record) else torch.from_numpy(np.asarray(record))
    return dict(boxes_3d=record[..., :10], scores_3d=record[..., 10], labels_3d=record[..., 11].long(),
                cls_scores=record[..., 12], instance_ids=lanes_to_ids(record[..., ID_LANES]))


def gather_detections(record, out=None, group=None):
    """all-gather of [streams, num_output, 15] records -> [world, streams, num_output, 15]."""
    if not (dist.is_available() and dist.is_initialized()):
        return record[None]
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(record.shape), dtype=record.dtype, device=record.device)
    # concatenation along dim 0 is the one output layout both RCCL and gloo accept
    dist.all_gather_into_tensor(out.view((-1,) + tuple(record.shape[1:])), record.contiguous(), group=group)
    return out


RECORD2D_WIDTH = 8   # box xyxy (4) + score + label + rank of its 3D box in the frame's record (-1: none) + camera (-1: pad)


def unpack_detections2d(record2d):
    """f32 [..., rows, 8] (the 2D device record csrc/decode.hip writes, padded with camera = -1 rows) -> list (one per
    leading index, flattened) of dicts boxes_2d / scores_2d / labels_2d / camidx_2d / rank3d of the rows that are associated
    with a kept 3D box: what decoder.py:176-251 puts into the result dict next to the 3D fields."""
    record2d = record2d.cpu() if torch.is_tensor(record2d) else torch.from_numpy(np.asarray(record2d))
    out = []
    for r in record2d.reshape(-1, record2d.shape[-2], record2d.shape[-1]):
        keep = (r[:, 7] >= 0) & (r[:, 6] >= 0)
        r = r[keep]
        out.append(dict(boxes_2d=r[:, :4], scores_2d=r[:, 4], labels_2d=r[:, 5].long(), camidx_2d=r[:, 7], rank3d=r[:, 6].long()))
    return out


def compact_record2d(rec2d, rows_out, out=None):
    """[bs, rows, 8] -> [bs, rows_out, 8]: the rows decode_with2d returns (rank >= 0 and camera >= 0) in their order, pad
    rows (zeros, rank -1, camera -1) behind. On the GPU one launch of csrc/decode.hip (simpb_record2d_compact); CPU tensors
    (the gloo tests of the exchange) take the same statement in torch."""
    bs, rows = rec2d.shape[:2]
    if out is None:
        out = torch.empty(bs, rows_out, RECORD2D_WIDTH, device=rec2d.device)
    if rec2d.is_cuda:
        from . import _lib
        from .plugin.ops import _ptr, _stream
        src = rec2d.contiguous().float()
        if out.dtype != torch.float32 or tuple(out.shape) != (bs, rows_out, RECORD2D_WIDTH) or out.stride(2) != 1 or out.stride(1) != RECORD2D_WIDTH:
            raise ValueError("compact_record2d: out must be f32 [bs, rows_out, 8] with dense rows (streams may be strided)")
        _lib.check(_lib.lib().simpb_record2d_compact(_ptr(out), out.stride(0) if bs > 1 else rows_out * RECORD2D_WIDTH, _ptr(src),
                                                     rows * RECORD2D_WIDTH, bs, rows, rows_out, _stream()), "simpb_record2d_compact")
        return out
    out.zero_()
    out[..., 6:8] = -1.0
    for b in range(bs):
        kept = rec2d[b][(rec2d[b, :, 6] >= 0) & (rec2d[b, :, 7] >= 0)][:rows_out]
        out[b, : kept.shape[0]] = kept
    return out


class DetectionGather:
    """The per-frame exchange: every rank's device records to every rank -- the 3D record [streams, num_output, 15] and,
    with rows2d > 0, the 2D record [streams, rows2d, 8] (decoder.py:230-251 returns boxes_2d / scores_2d / labels_2d /
    camidx_2d beside the 3D fields; apis/test.py:49-119 collects the whole dict) -- in ONE all_gather_into_tensor per
    frame, off the compute streams. rows2d is a fixed exchange capacity equal on every rank:
    * compact2d=True (what bench.py uses): rows2d = num_output x num_cams and a runner's record is COMPACTED on the device
      to the rows decode_with2d returns (slots of the kept 3D boxes, at most one per box and camera) -- 1 800 rows for 300
      boxes x 6 cameras whatever the runners' slot capacities are, so a rank whose 2D set overflows and grows its capacity
      mid-stream (runner.py) changes nothing about the exchange, and the ranks never have to agree on a size;
    * compact2d=False: the slot array itself, rows2d = num_anchor x num_cams (5 400: no set can exceed it); a runner's
      record is copied into the leading rows, the rest stay pad rows (camera = -1).

    The send and receive buffers are persistent and owned by the side stream, so no allocator block crosses streams;
    the records handed to submit() are read on the side stream after it has waited for their producer streams and are
    marked with record_stream, so the caching allocator cannot hand their blocks out while the copy is pending. Records
    that live in a replayed graph's memory are protected by the runners: they make every graph that may write that memory
    wait for `done` (runner.rec_consumed). The host never waits inside submit(); result() waits for the last exchange."""

    def __init__(self, streams, num_output, device, group=None, rows2d=0, compact2d=False):
        self.device = torch.device(device)
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.n3, self.rows2d = num_output * RECORD_WIDTH, int(rows2d)
        self.compact2d = bool(compact2d) and self.rows2d > 0
        width = self.n3 + self.rows2d * RECORD2D_WIDTH
        self.send_buf = torch.zeros(streams, width, device=self.device)
        self.recv_buf = torch.zeros(self.world, streams, width, device=self.device)
        self.send = self.send_buf[:, :self.n3].unflatten(1, (num_output, RECORD_WIDTH))
        self.recv = self.recv_buf[..., :self.n3].unflatten(2, (num_output, RECORD_WIDTH))
        self.send2d = self.send_buf[:, self.n3:].unflatten(1, (self.rows2d, RECORD2D_WIDTH)) if self.rows2d else None
        self.recv2d = self.recv_buf[..., self.n3:].unflatten(2, (self.rows2d, RECORD2D_WIDTH)) if self.rows2d else None
        if self.rows2d:
            self.send2d[..., 6:8] = -1.0   # pad rows: no 3D partner, no camera
        self.done = None
        self.frames = 0
        # rehearsal of N > 1 on one GPU box (bench.py --backend gloo): gloo has no device all-gather, stage through the host
        self.via_host = self.cuda and self.world > 1 and dist.get_backend(group) == "gloo"

    def _exchange(self, records, records2d):
        at = 0
        for r in records:
            n = r.shape[0]
            if r.shape[1:] != self.send.shape[1:]:
                raise ValueError(f"record {tuple(r.shape)} does not match {tuple(self.send.shape)}")
            self.send[at:at + n].copy_(r, non_blocking=True)
            at += n
        if at != self.send.shape[0]:
            raise ValueError(f"{at} stream records submitted, {self.send.shape[0]} expected")
        if self.rows2d:
            if records2d is None:
                raise ValueError("this exchange carries the 2D records as well: pass records2d")
            at = 0
            for r in records2d:
                n, rows = r.shape[:2]
                if r.shape[2] != RECORD2D_WIDTH:
                    raise ValueError(f"2D record {tuple(r.shape)} is not [*, rows, {RECORD2D_WIDTH}]")
                if self.compact2d:   # only the rows of the kept 3D boxes, straight into the send buffer
                    compact_record2d(r, self.rows2d, out=self.send2d[at:at + n])
                    at += n
                    continue
                # a runner's slot array may be LONGER than the exchange capacity (capacities are rounded up to 128: 5 504 for
                # 900 x 6), never its live part: live slots come first and there are at most num_anchor x num_cams of them
                rows = min(rows, self.rows2d)
                self.send2d[at:at + n, :rows].copy_(r[:, :rows], non_blocking=True)   # (rows past it are pad rows and stay so: capacities only grow)
                at += n
            if at != self.send.shape[0]:
                raise ValueError(f"{at} 2D stream records submitted, {self.send.shape[0]} expected")
        flat = self.recv_buf.view(-1, self.recv_buf.shape[-1])
        if self.world > 1 and self.via_host:
            host = torch.empty(flat.shape, dtype=flat.dtype)
            dist.all_gather_into_tensor(host, self.send_buf.cpu(), group=self.group)
            flat.copy_(host)
        elif self.world > 1:
            dist.all_gather_into_tensor(flat, self.send_buf, group=self.group)
        else:
            self.recv_buf[0].copy_(self.send_buf, non_blocking=True)

    def submit(self, records, producers=(), records2d=None):
        """records: device tensors [bs_i, num_output, 15] of this rank's runners, in stream order (records2d: their 2D
        records [bs_i, rows_i, 8]); producers: the streams they were written on (the side stream waits for them; the
        current stream is always waited for)."""
        if not self.cuda:
            self._exchange(records, records2d)
            self.frames += 1
            return
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        for s in producers:
            self.side.wait_stream(s)
        with torch.cuda.stream(self.side):
            self._exchange(records, records2d)
            self.done = torch.cuda.Event()
            self.done.record(self.side)
        for r in list(records) + list(records2d or ()):
            r.record_stream(self.side)
        self.frames += 1

    def result(self):
        """[world, streams, num_output, 15] of the last submitted frame (waits for that exchange)."""
        if self.done is not None:
            self.done.synchronize()
        return self.recv

    def result2d(self):
        """[world, streams, rows2d, 8] of the last submitted frame (pad rows: camera = -1)."""
        if self.done is not None:
            self.done.synchronize()
        return self.recv2d
