"""Host-side scheduling logic of the pipelined runner (no GPU): which steps may overlap the two streams."""
from simpb_amd.runner import PipelinedRunner


def _bare(**kw):
    r = object.__new__(PipelinedRunner)
    base = dict(use_graph=True, bb_graph=[None, None], head_graph=[None, None], bb_out=[None, None], fm=[None, None],
                pending=None, prev_metas=None)
    base.update(kw)
    for k, v in base.items():
        setattr(r, k, v)
    return r


def test_warm_up_steps_are_exclusive():
    assert PipelinedRunner.SERIALIZE_EAGER is True
    # nothing captured yet
    assert not _bare()._replay_only(0, False)
    # backbone graph exists for the slot, no decoder pending: a lone replay
    maps = object()
    r = _bare(bb_graph=["g", None], bb_out=[maps, None], fm=[maps, None])
    assert r._replay_only(0, False)
    assert not r._replay_only(0, True)       # forced eager step (bench.py's metering leg)
    assert not r._replay_only(1, False)      # the other slot still runs eagerly
    # decoder pending on slot 1 without its graph -> not overlapped
    r = _bare(bb_graph=["g0", "g1"], bb_out=[maps, maps], fm=[maps, maps], pending=(1, {}), prev_metas={})
    assert not r._replay_only(0, False)
    # all four graphs in place and the pending features are the captured buffer -> overlap
    r.head_graph = ["h0", "h1"]
    assert r._replay_only(0, False)
    # features of the pending slot produced eagerly (not the captured buffer) -> its decoder cannot replay
    r.fm = [maps, object()]
    assert not r._replay_only(0, False)
    # cold decoder (no previous frame) never replays
    r.fm = [maps, maps]
    r.prev_metas = None
    assert not r._replay_only(0, False)
    # graphs disabled
    assert not _bare(use_graph=False, bb_graph=["g", None])._replay_only(0, False)
