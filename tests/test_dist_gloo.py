"""CPU, world_size 2, gloo: the stream sharding and the detection all-gather that bench.py runs
over RCCL on the GPU box."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simpb_amd.dist import RECORD_WIDTH, gather_detections, pack_detections, shard_streams


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_results(stream_ids):
    out = []
    for s in stream_ids:
        g = torch.Generator().manual_seed(100 + s)
        out.append({"img_bbox": dict(boxes_3d=torch.randn(300, 10, generator=g), scores_3d=torch.rand(300, generator=g),
                                     labels_3d=torch.randint(0, 10, (300,), generator=g),
                                     cls_scores=torch.rand(300, generator=g),
                                     instance_ids=torch.arange(300) + 1000 * s)})
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(4, rank, world)
    rec = pack_detections(_fake_results(mine))
    allrec = gather_detections(rec)
    q.put((rank, mine, allrec.numpy().copy()))  # by value: the worker may exit before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_shard_streams_partition():
    for n in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            got = sum((shard_streams(n, r, world) for r in range(world)), [])
            assert got == list(range(n))


def test_gather_detections_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == [0, 1] and got[1][1] == [2, 3]
    want = torch.stack([pack_detections(_fake_results([0, 1])), pack_detections(_fake_results([2, 3]))])
    for _, _, allrec in got:
        assert allrec.shape == (2, 2, 300, RECORD_WIDTH)
        assert torch.equal(torch.from_numpy(allrec), want)
