"""-m gpu: the fusion group -- several row-sharded training calls of a rank in flight at once, ONE collective per step for all of them
(include/rgbm.h "Fusion group"; the reference trains its targets in parallel, python/repair/model.py:817-926).

The box has one GPU, so the ranks are host threads on it (thread-group transport) or a single RCCL rank (world of one); every rank runs
its own fusion group with one member thread per target.  Every model must be the single-device model bit for bit, whatever the members'
collective sequences look like next to each other (a binary target next to a K = 12 one, a member that trains two targets while the others
train one, bagging, different numbers of boosting iterations), and a failing member must fail the others instead of hanging them."""
import threading

import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


def _kw(cards, t, **over):
    K = int(cards[t])
    return dict(dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=6, learning_rate=0.2), **over)


def _fused_rank(N, group_ctx, shard_table, plan, dirty, cards, out, err, key):
    """One rank: a fusion group with one member per entry of `plan` (a list of target lists), every member on its own thread."""
    fg = N.FusionGroup(len(plan))

    def member(j):
        try:
            with fg.member(j):
                for t, over in plan[j]:
                    feats = [c for c in range(dirty.shape[0]) if c != t]
                    out[(key, t)] = shard_table.train(t, feats, class_weight=balanced_weights(dirty[t], int(cards[t])), row_sharded=True, **_kw(cards, t, **over)).save()
        except Exception as e:  # noqa: BLE001
            err.append((key, j, e))

    ths = [threading.Thread(target=member, args=(j,)) for j in range(len(plan))]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=600)
    info = fg.info()
    fg.close()
    return info


def _run(dirty, cards, bounds, plan):
    from repair import _native as N
    nr = len(bounds) - 1
    lg = N.LocalGroup(nr)
    out, err, infos = {}, [], [None] * nr

    def rank(r):
        try:
            lg.join(r)
            try:
                tab = N.Table(np.ascontiguousarray(dirty[:, bounds[r]:bounds[r + 1]]), cards)
                infos[r] = _fused_rank(N, lg, tab, plan, dirty, cards, out, err, r)
            finally:
                N.comm_finalize()
        except Exception as e:  # noqa: BLE001
            err.append((r, -1, e))

    ths = [threading.Thread(target=rank, args=(r,)) for r in range(nr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=900)
    return out, err, infos


def test_three_targets_in_flight_on_two_thread_ranks_give_the_single_device_models():
    from repair import _native as N
    dirty, _, cards = make_table(30000, 8, seed=77, null_ratio=0.02)
    # member 0: the binary target, then the K = 3 one (two training calls: its sequence is twice as long); member 1: K = 12 with bagging
    # (two more collectives at set-up, one per bagging round); member 2: K = 8 with fewer iterations (it leaves while the others go on)
    plan = [[(0, {}), (1, {})], [(5, dict(bagging_fraction=0.7, bagging_freq=2))], [(4, dict(n_estimators=3))]]
    out, err, infos = _run(dirty, cards, [0, 13000, 30000], plan)
    assert not err, err
    full = N.Table(dirty, cards)
    for members in plan:
        for t, over in members:
            feats = [c for c in range(8) if c != t]
            single = full.train(t, feats, class_weight=balanced_weights(dirty[t], int(cards[t])), **_kw(cards, t, **over)).save()
            for r in range(2):
                assert out[(r, t)] == single, "rank %d, target c%d: the fused row-sharded model differs from the single-device model" % (r, t)
    # one all-reduce per step and element type for all members: far fewer collectives than member parts carried
    assert infos[0]["collectives"] == infos[1]["collectives"] and infos[0]["parts"] == infos[1]["parts"]
    assert infos[0]["parts"] > infos[0]["collectives"] and infos[0]["members_in"] == 0 and not infos[0]["broken"]


def test_fusion_group_over_an_rccl_world_of_one():
    from repair import _native as N
    dirty, _, cards = make_table(20000, 8, seed=78, null_ratio=0.02)
    N.comm_init(N.comm_unique_id(), 0, 1, 0)
    try:
        tab = N.Table(dirty, cards)
        out, err = {}, []
        plan = [[(4, {})], [(0, {})], [(6, {})]]
        info = _fused_rank(N, None, tab, plan, dirty, cards, out, err, 0)
        assert not err, err
        assert N.comm_info()["kind"] == 1, "the communicator must be back on the calling thread"
        for members in plan:
            for t, over in members:
                feats = [c for c in range(8) if c != t]
                assert out[(0, t)] == tab.train(t, feats, class_weight=balanced_weights(dirty[t], int(cards[t])), **_kw(cards, t, **over)).save()
        assert info["parts"] >= 3 * info["collectives"] - 6 and not info["broken"]
    finally:
        N.comm_finalize()


def test_a_failing_member_fails_the_others_instead_of_hanging_them():
    from repair import _native as N
    dirty, _, cards = make_table(20000, 8, seed=79, null_ratio=0.02)
    # member 1 asks for a binary objective on a 12-class target: its call fails during set-up, after its first collective
    plan = [[(4, dict(n_estimators=40))], [(5, dict(objective=0, num_class=2))]]
    out, err, infos = _run(dirty, cards, [0, 20000], plan)
    assert len(err) == 2, err
    msgs = " | ".join(str(e[2]) for e in err)
    assert "fusion group" in msgs
    assert (0, 4) not in out
