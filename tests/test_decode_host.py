"""CPU: the host half of the static decode (plugin/detection3d.py:decode_static_host) on hand-made records -- the
reference's per-sample dict (decoder.py:176-251) for a batch with the reference's semantics (the group table of sample 0
carried along the batch, :216) and for a batch of independent streams (each record decoded as the batch of one it is)."""
import numpy as np

from simpb_amd.plugin.detection3d import SparseBox3DDecoder


def _records(seed, live, rows=40, k=12):
    """One stream: rec3d [k, 15]; rec2d [rows, 8] whose first `live` rows are camera-major slots, the rest pad rows."""
    rng = np.random.default_rng(seed)
    rec3d = rng.standard_normal((k, 15)).astype(np.float32)
    rec2d = np.zeros((rows, 8), np.float32)
    rec2d[:, 6:8] = -1.0
    cams = np.sort(rng.integers(0, 6, size=live))
    rec2d[:live, :6] = rng.standard_normal((live, 6)).astype(np.float32)
    rec2d[:live, 7] = cams
    partner = rng.permutation(k)[: min(k, live)]
    has = rng.random(live) < 0.6
    rec2d[:live, 6] = np.where(has, np.resize(partner, live), -1)
    return rec3d, rec2d


def test_independent_streams_are_decoded_one_by_one():
    recs = [_records(s, live) for s, live in enumerate((17, 5, 0, 31))]
    rec3d, rec2d = np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs])
    got = SparseBox3DDecoder.decode_static_host(rec3d, rec2d, 6, independent=True)
    assert len(got) == 4
    for b, (r3, r2) in enumerate(recs):
        want = SparseBox3DDecoder.decode_static_host(r3[None], r2[None], 6)[0]
        assert set(got[b]) == set(want)
        for key in want:
            if key == "query_groups":
                assert got[b][key] == want[key]
            else:
                assert np.array_equal(np.asarray(got[b][key]), np.asarray(want[key])), (b, key)
        kept = (r2[:, 6] >= 0) & (r2[:, 7] >= 0)
        assert len(want["boxes_2d"]) == int(kept.sum()) and tuple(want["trans_matrix"].shape) == (12, int(kept.sum()))
        assert np.array_equal(np.asarray(want["trans_matrix"]).sum(0), np.ones(int(kept.sum()), np.float32))


def test_reference_batch_carries_the_group_table_of_sample_zero():
    """decoder.py:216 re-binds the group list inside the loop over the batch: sample i > 0 is bucketed with the groups sample
    i - 1 produced. The default (independent=False) keeps that; it only shows when the samples' groups differ."""
    recs = [_records(s, live) for s, live in enumerate((17, 9))]
    rec3d, rec2d = np.stack([r[0] for r in recs]), np.stack([r[1] for r in recs])
    both = SparseBox3DDecoder.decode_static_host(rec3d, rec2d, 6)
    alone = SparseBox3DDecoder.decode_static_host(rec3d[1:], rec2d[1:], 6)[0]
    assert np.array_equal(np.asarray(both[1]["boxes_2d"]), np.asarray(alone["boxes_2d"]))        # the rows themselves do not depend on it
    assert both[0]["query_groups"] == SparseBox3DDecoder.decode_static_host(rec3d[:1], rec2d[:1], 6)[0]["query_groups"]
    assert both[1]["query_groups"] != alone["query_groups"] or len(both[1]["camidx_2d"]) != len(alone["camidx_2d"])
