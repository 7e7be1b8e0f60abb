"""Multi-GPU layer of the path: camera streams are independent (frame t of a
stream reads the bank frame t-1 of the SAME stream wrote), so streams are sharded across ranks with
no collective inside a frame; the one exchange is an all-gather of a fixed-shape detection record
per frame (backend 'nccl' = RCCL over xGMI on the GPU box, 'gloo' in the CPU tests). Replaces the
reference's pickle-to-tmpdir + barrier collection (apis/test.py:122-171)."""
import torch
import torch.distributed as dist

RECORD_WIDTH = 14  # 10 box + score + label + cls_score + instance id (decoder.py:230-251)


def shard_streams(num_streams, rank, world):
    """Contiguous block of stream ids for this rank (the reference shards whole scenes
    contiguously per rank: datasets/samplers/distributed_sampler.py:61-79)."""
    per, rem = divmod(num_streams, world)
    start = rank * per + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))


def pack_detections(results, device=None, num_output=300):
    """list of per-stream result dicts (head.post_process) -> f32 [streams, num_output, 14];
    rows past a stream's own count (score_threshold set) are zero with id -1."""
    recs = []
    for r in results:
        d = r["img_bbox"] if "img_bbox" in r else r
        n = d["boxes_3d"].shape[0]
        rec = torch.zeros(num_output, RECORD_WIDTH)
        rec[:, 13] = -1
        rec[:n, :10] = d["boxes_3d"].cpu()
        rec[:n, 10] = d["scores_3d"].cpu()
        rec[:n, 11] = d["labels_3d"].cpu().float()
        rec[:n, 12] = d["cls_scores"].cpu()
        rec[:n, 13] = d["instance_ids"].cpu().float()
        recs.append(rec)
    out = torch.stack(recs)
    return out if device is None else out.to(device, non_blocking=True)


def gather_detections(record, out=None, group=None):
    """all-gather of [streams, num_output, 14] records -> [world, streams, num_output, 14]."""
    if not (dist.is_available() and dist.is_initialized()):
        return record[None]
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(record.shape), dtype=record.dtype, device=record.device)
    # concatenation along dim 0 is the one output layout both RCCL and gloo accept
    dist.all_gather_into_tensor(out.view((-1,) + tuple(record.shape[1:])), record.contiguous(), group=group)
    return out
