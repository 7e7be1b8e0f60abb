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


class DetectionGather:
    """The per-frame exchange: every rank's device records [streams, num_output, 15] to every rank, one
    all_gather_into_tensor per frame, off the compute streams.

    The send and receive buffers are persistent and owned by the side stream, so no allocator block crosses streams;
    the records handed to submit() are read on the side stream after it has waited for their producer streams and are
    marked with record_stream, so the caching allocator cannot hand their blocks out while the copy is pending. The
    host never waits inside submit(); result() waits for the last exchange only."""

    def __init__(self, streams, num_output, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.cuda = self.device.type == "cuda"
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.send = torch.zeros(streams, num_output, RECORD_WIDTH, device=self.device)
        self.recv = torch.zeros(self.world, streams, num_output, RECORD_WIDTH, device=self.device)
        self.done = None
        self.frames = 0
        # rehearsal of N > 1 on one GPU box (bench.py --backend gloo): gloo has no device all-gather, stage through the host
        self.via_host = self.cuda and self.world > 1 and dist.get_backend(group) == "gloo"

    def _exchange(self, records):
        at = 0
        for r in records:
            n = r.shape[0]
            if r.shape[1:] != self.send.shape[1:]:
                raise ValueError(f"record {tuple(r.shape)} does not match {tuple(self.send.shape)}")
            self.send[at:at + n].copy_(r, non_blocking=True)
            at += n
        if at != self.send.shape[0]:
            raise ValueError(f"{at} stream records submitted, {self.send.shape[0]} expected")
        if self.world > 1 and self.via_host:
            host = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(host.view((-1,) + tuple(self.send.shape[1:])), self.send.cpu(), group=self.group)
            self.recv.copy_(host)
        elif self.world > 1:
            dist.all_gather_into_tensor(self.recv.view((-1,) + tuple(self.send.shape[1:])), self.send, group=self.group)
        else:
            self.recv[0].copy_(self.send, non_blocking=True)

    def submit(self, records, producers=()):
        """records: device tensors [bs_i, num_output, 15] of this rank's runners, in stream order; producers: the
        streams they were written on (the side stream waits for them; the current stream is always waited for)."""
        if not self.cuda:
            self._exchange(records)
            self.frames += 1
            return
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        for s in producers:
            self.side.wait_stream(s)
        with torch.cuda.stream(self.side):
            self._exchange(records)
            self.done = torch.cuda.Event()
            self.done.record(self.side)
        for r in records:
            r.record_stream(self.side)
        self.frames += 1

    def result(self):
        """[world, streams, num_output, 15] of the last submitted frame (waits for that exchange)."""
        if self.done is not None:
            self.done.synchronize()
        return self.recv
