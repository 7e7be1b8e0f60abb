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


# ---------------------------------------------------------------------------------------------------------------------
# The exchange bench.py uses since round 4: compact2d=True -- only the 2D rows of the kept 3D boxes travel, compacted into a
# fixed num_output x num_cams rows, so the exchange shape does not depend on any runner's slot capacity.
COMPACT_ROWS = 64


def _runner_record2d(stream, frame, rows):
    """A runner's 2D record with `rows` slots (its static capacity at that frame): live slots first, a third of them
    belonging to kept 3D boxes (rank >= 0), capacity slots (camera -1) behind."""
    g = torch.Generator().manual_seed(991 * frame + stream)
    live = 30 + (frame * 7 + stream * 3) % 12
    rec = torch.zeros(1, rows, RECORD2D_WIDTH)
    rec[0, :, 6:8] = -1.0
    rec[0, :live, :5] = torch.rand(live, 5, generator=g)
    rec[0, :live, 5] = torch.randint(0, 10, (live,), generator=g).float()
    rank = torch.randint(0, 300, (live,), generator=g).float()
    rank[torch.rand(live, generator=g) > 0.35] = -1.0
    rec[0, :live, 6] = rank
    rec[0, :live, 7] = torch.sort(torch.randint(0, 6, (live,), generator=g)).values.float()
    return rec


def _capacity(rank, frame):
    """Rank 1's runner overflows at frame 3: it re-runs the frame at a larger capacity and re-captures its graphs
    (runner.py), so from then on its 2D record has 2304 rows instead of 1536 / 48; rank 0 never changes."""
    return 48 if rank == 0 or frame < 3 else 176


def _overflow_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_streams(2, rank, world)
    gather = DetectionGather(len(mine), 300, torch.device("cpu"), rows2d=COMPACT_ROWS, compact2d=True)
    seen = []
    for frame in range(6):
        if rank == 1 and frame == 3:
            time.sleep(0.3)   # the re-run + re-capture happen INSIDE the runner's step: one submit per step, only later
        gather.submit(_device_records(mine, frame), records2d=[_runner_record2d(s, frame, _capacity(rank, frame)) for s in mine])
        seen.append((gather.result().numpy().view("int32").copy(), gather.result2d().numpy().copy()))
    q.put((rank, gather.frames, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_compact_exchange_is_untouched_by_a_rank_whose_runner_overflows_mid_stream():
    """World 2 over gloo: rank 1's runner overflows at frame 3 (its step takes longer and its 2D record grows from 48 to
    176 rows) while rank 0 keeps submitting. The exchange carries the compacted 2D rows, so its shape never changes: six
    steps = six matched collectives on both ranks, and every frame's gathered 3D and 2D records are the runners' records
    (3D bit for bit, 2D = the rows of the kept boxes in slot order, pad rows behind) on both ranks."""
    from simpb_amd.dist import compact_record2d
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, frames, seen in got:
        assert frames == 6 and len(seen) == 6
        for frame, (rec3d, rec2d) in enumerate(seen):
            assert rec3d.shape == (2, 1, 300, RECORD_WIDTH) and rec2d.shape == (2, 1, COMPACT_ROWS, RECORD2D_WIDTH)
            for src in range(2):   # the record of stream `src` (owned by rank `src`) as every rank holds it
                want3d = _device_records([src], frame)[0]
                assert (rec3d[src] == want3d.numpy().view("int32")).all()
                full = _runner_record2d(src, frame, _capacity(src, frame))
                want2d = compact_record2d(full, COMPACT_ROWS)
                assert torch.equal(torch.from_numpy(rec2d[src]), want2d)
                kept = full[0][(full[0, :, 6] >= 0) & (full[0, :, 7] >= 0)]
                a, b = unpack_detections2d(torch.from_numpy(rec2d[src]))[0], unpack_detections2d(full)[0]
                assert len(kept) > 3 and torch.equal(a["boxes_2d"], b["boxes_2d"]) and torch.equal(a["rank3d"], b["rank3d"])
                assert torch.equal(a["camidx_2d"], b["camidx_2d"])
