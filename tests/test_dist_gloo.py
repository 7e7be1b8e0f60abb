"""CPU, world_size 2, gloo: the stream sharding and the detection all-gather that bench.py runs
over RCCL on the GPU box."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simpb_amd.dist import (RECORD2D_WIDTH, RECORD_WIDTH, DetectionGather, gather_detections, ids_to_lanes, lanes_to_ids,
                            pack_detections, shard_streams, unpack_detections, unpack_detections2d)


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
                                     instance_ids=torch.arange(300) + 1000 * s + (1 << 33))})  # far above 2^24
    return out


def _device_records(stream_ids, frame):
    """What the runners hand to the exchange: one [1, 300, 15] record per stream in the layout csrc/decode.hip writes
    (ids as two bit-cast lanes; -1 ids are NaN bit patterns as floats, so everything is compared through int views)."""
    recs = []
    for s in stream_ids:
        g = torch.Generator().manual_seed(1000 * frame + s)
        rec = torch.randn(1, 300, RECORD_WIDTH, generator=g)
        ids = torch.arange(300) + 300 * frame + (1 << 25) * (s + 1)
        ids[::7] = -1
        rec[0, :, 13:15] = ids_to_lanes(ids)
        recs.append(rec)
    return recs


ROWS2D = 96   # exchange capacity of the 2D record in these tests (bench.py: num_anchor x num_cams)


def _device_records2d(stream_ids, frame):
    """The runners' 2D records: [1, rows, 8] with rows that differ per stream and frame (a runner's static capacity can
    grow), camera-major slots, some rows without a 3D partner (rank -1) and capacity slots (camera -1)."""
    recs = []
    for s in stream_ids:
        g = torch.Generator().manual_seed(77 * frame + s)
        rows = 40 + 8 * min(frame + s % 2, 2)   # never shrinks: a runner's capacity only grows
        rec = torch.rand(1, rows, RECORD2D_WIDTH, generator=g)
        rec[0, :, 5] = torch.randint(0, 10, (rows,), generator=g).float()
        rec[0, :, 6] = torch.randint(-1, 300, (rows,), generator=g).float()
        rec[0, :, 7] = torch.sort(torch.randint(0, 6, (rows,), generator=g)).values.float()
        rec[0, rows - 5:, 6:8] = -1.0
        recs.append(rec)
    return recs


def _gather_worker(rank, world, port, q):
    """bench.py's step(): every frame the rank's runners' records go through DetectionGather.submit, the next frame
    is submitted without waiting, result() is read at the end (and once in the middle)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(4, rank, world)
    gather = DetectionGather(len(mine), 300, torch.device("cpu"), rows2d=ROWS2D)
    seen, seen2d = [], []
    for frame in range(3):
        gather.submit(_device_records(mine, frame), records2d=_device_records2d(mine, frame))
        if frame == 1:
            seen.append(gather.result().clone())
            seen2d.append(gather.result2d().clone())
    seen.append(gather.result().clone())
    seen2d.append(gather.result2d().clone())
    q.put((rank, [x.numpy().view("int32").copy() for x in seen], gather.frames, [x.numpy().copy() for x in seen2d]))
    dist.barrier()
    dist.destroy_process_group()


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
        assert (allrec.view("int32") == want.numpy().view("int32")).all()


def test_ids_travel_bit_exactly():
    ids = torch.tensor([-1, 0, (1 << 24) + 1, (1 << 31) + 5, (1 << 40) + 3, -(1 << 35)])
    lanes = ids_to_lanes(ids)
    assert lanes.dtype == torch.float32 and lanes.shape == (6, 2)
    assert torch.equal(lanes_to_ids(lanes), ids) and (lanes_to_ids(lanes.numpy()) == ids.numpy()).all()
    rec = pack_detections(_fake_results([3]))
    got = unpack_detections(rec)
    assert got["instance_ids"].dtype == torch.int64
    assert torch.equal(got["instance_ids"][0], torch.arange(300) + 3000 + (1 << 33))


def test_detection_gather_world2_drives_the_bench_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for frame, which in ((1, 0), (2, 1)):
        want = torch.stack([torch.cat(_device_records([0, 1], frame)), torch.cat(_device_records([2, 3], frame))])
        for _, seen, frames, seen2d in got:
            assert frames == 3 and seen[which].shape == (2, 2, 300, RECORD_WIDTH)
            assert (seen[which] == want.numpy().view("int32")).all()  # every rank holds every stream's record, bit for bit
            # ... and every stream's 2D record: its own rows in front, pad rows (camera -1, no 3D partner) behind
            assert seen2d[which].shape == (2, 2, ROWS2D, RECORD2D_WIDTH)
            for s in range(4):
                rec = _device_records2d([s], frame)[0][0]
                have = torch.from_numpy(seen2d[which][s // 2, s % 2])
                assert torch.equal(have[: rec.shape[0]], rec) and bool((have[rec.shape[0]:, 6:8] == -1).all())
                keep = (rec[:, 7] >= 0) & (rec[:, 6] >= 0)
                got2d = unpack_detections2d(have)[0]
                assert torch.equal(got2d["boxes_2d"], rec[keep, :4]) and torch.equal(got2d["camidx_2d"], rec[keep, 7])
        ids = unpack_detections(torch.from_numpy(got[0][1][which].view("float32")))["instance_ids"]
        assert int(ids[1, 0, 1]) == 1 + 300 * frame + (1 << 25) * 3 and int(ids[0, 0, 0]) == -1


def test_detection_gather_takes_a_slot_array_longer_than_the_exchange_capacity():
    """A runner's capacity is rounded up to a multiple of 128 (5 504 slots for 900 anchors x 6 cameras) while the exchange
    carries num_anchor x num_cams = 5 400 rows: live slots come first and can never exceed that, so the surplus rows are pad
    rows and are simply not sent (world 1, CPU)."""
    from simpb_amd.dist import DetectionGather, unpack_detections2d
    rows2d = 24
    g = DetectionGather(2, 300, torch.device("cpu"), rows2d=rows2d)
    rec3d = torch.randn(2, 300, 15)
    rec2d = torch.zeros(2, rows2d + 8, 8)
    rec2d[..., 6:8] = -1.0
    rec2d[0, :5, :6] = torch.randn(5, 6)
    rec2d[0, :5, 6] = torch.arange(5.0)
    rec2d[0, :5, 7] = 2.0
    g.submit([rec3d], records2d=[rec2d])
    out = g.result2d()
    assert tuple(out.shape) == (1, 2, rows2d, 8) and torch.equal(out[0], rec2d[:, :rows2d])
    got = unpack_detections2d(out[0])
    assert len(got[0]["boxes_2d"]) == 5 and len(got[1]["boxes_2d"]) == 0
