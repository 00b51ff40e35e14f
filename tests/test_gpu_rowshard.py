"""-m gpu: row-sharded (data-parallel) training == single-device training, bit for bit.

Every rank holds a row shard and the ranks exchange integer all-reduces only (code counts, per-level
histograms, child counts), so the model must not depend on the number of ranks or on the split.
A real multi-GPU run needs one process per GPU (RCCL); the box the tests run on has ONE GPU, so the
sharding logic is exercised with the in-process thread-group transport (one rank per host thread on
the same device) and the RCCL transport with a world of one.
"""
import threading

import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["default", "one_tree_small_lds"], autouse=True)
def level_pass_kind(request, monkeypatch):
    """Every case runs with the level pass as configured by default and with one class tree per workgroup + a small LDS pool (several
    launches per level)."""
    if request.param == "one_tree_small_lds":
        monkeypatch.setenv("RGBM_MT_TREES", "1")
        monkeypatch.setenv("RGBM_LV_LDS", "99000")
    return request.param


def _train_sharded(dirty, cards, bounds, target, feats, cw, kw):
    from repair import _native as N
    nr = len(bounds) - 1
    group = N.LocalGroup(nr)
    out, err = [None] * nr, [None] * nr

    def work(r):
        try:
            group.join(r)
            try:
                tab = N.Table(np.ascontiguousarray(dirty[:, bounds[r]:bounds[r + 1]]), cards)
                out[r] = tab.train(target, feats, class_weight=cw, row_sharded=True, **kw).save()
            finally:
                N.comm_finalize()
        except Exception as e:  # noqa: BLE001
            err[r] = e

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("target,bounds", [(4, [0, 9000, 20000]), (5, [0, 3000, 3500, 14000, 20000]), (0, [0, 1, 20000])])
def test_thread_group_shards_give_the_single_device_model(target, bounds):
    from repair import _native as N
    dirty, clean, cards = make_table(20000, 8, seed=71, null_ratio=0.02)
    feats = [c for c in range(8) if c != target]
    K = int(cards[target])
    cw = balanced_weights(dirty[target], K)
    kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=8, learning_rate=0.2)
    single = N.Table(dirty, cards).train(target, feats, class_weight=cw, **kw).save()
    shards = _train_sharded(dirty, cards, bounds, target, feats, cw, kw)
    for r, b in enumerate(shards):
        assert b == single, "rank %d of %d differs from the single-device model" % (r, len(shards))


def test_two_chunks_and_many_bins_sharded():
    from repair import _native as N
    rng = np.random.default_rng(73)
    n = 24000
    z = rng.integers(0, 120, n)
    X = np.stack([((z * (j + 1) + rng.integers(0, 9, n)) % (20 + 11 * j)).astype(np.int32) for j in range(19)])
    y = ((z // 10 + X[3] % 3) % 5).astype(np.int32)
    tab = np.ascontiguousarray(np.vstack([X, y[None, :]]))
    cards = np.asarray([20 + 11 * j for j in range(19)] + [5], np.int32)
    cw = balanced_weights(y, 5)
    kw = dict(objective=1, num_class=5, n_estimators=4, learning_rate=0.3)
    feats = list(range(19))
    single = N.Table(tab, cards).train(19, feats, class_weight=cw, **kw).save()
    for b in _train_sharded(tab, cards, [0, 7000, 15000, 24000], 19, feats, cw, kw):
        assert b == single


def test_rccl_world_of_one():
    from repair import _native as N
    dirty, clean, cards = make_table(12000, 6, seed=79)
    t = 3
    feats = [c for c in range(6) if c != t]
    K = int(cards[t])
    cw = balanced_weights(dirty[t], K)
    kw = dict(objective=1, num_class=K, n_estimators=5, learning_rate=0.2)
    single = N.Table(dirty, cards).train(t, feats, class_weight=cw, **kw).save()
    N.comm_init(N.comm_unique_id(), 0, 1, 0)
    try:
        assert N.comm_info() == dict(kind=1, rank=0, nranks=1)
        sharded = N.Table(dirty, cards).train(t, feats, class_weight=cw, row_sharded=True, **kw).save()
        local = N.Table(dirty, cards).train(t, feats, class_weight=cw, **kw).save()   # same thread, flag off: plain local training
        assert local == single
    finally:
        N.comm_finalize()
    assert sharded == single
    assert N.comm_info()["kind"] == 0


def test_row_sharded_rejects_the_leafwise_grower():
    from repair import _native as N
    dirty, clean, cards = make_table(4000, 5, seed=83)
    group = N.LocalGroup(1)
    group.join(0)
    try:
        tab = N.Table(dirty, cards)
        with pytest.raises(N.RepairGbmError):
            tab.train(1, [0, 2, 3, 4], objective=1, num_class=int(cards[1]), n_estimators=2, max_depth=-1, row_sharded=True)
    finally:
        N.comm_finalize()


@pytest.mark.parametrize("target,bounds,kw", [
    (4, [0, 9000, 20000], dict(bagging_fraction=0.6, bagging_freq=2)),
    (5, [0, 3000, 3500, 14000, 20000], dict(bagging_fraction=0.5, bagging_freq=1, feature_fraction=0.7)),   # shard borders inside a 1024-position block
    (0, [0, 1, 20000], dict(bagging_fraction=0.8, bagging_freq=3)),
    (2, [0, 700, 20000], dict(bagging_fraction=0.55, bagging_freq=1)),                                       # a shard smaller than one block
])
def test_bagging_under_row_sharding(target, bounds, kw):
    """GBDT::Bagging draws per training-row position over the WHOLE table: every rank walks the LCG blocks that overlap its positions
    and the bag size is summed over the ranks -- same model as on one device (and, by the grower tests, as the oracle)."""
    from repair import _native as N
    dirty, clean, cards = make_table(20000, 8, seed=89, null_ratio=0.03)
    feats = [c for c in range(8) if c != target]
    K = int(cards[target])
    cw = balanced_weights(dirty[target], K)
    kw = dict(kw, objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=7, learning_rate=0.2)
    single = N.Table(dirty, cards).train(target, feats, class_weight=cw, **kw).save()
    for r, b in enumerate(_train_sharded(dirty, cards, bounds, target, feats, cw, kw)):
        assert b == single, "rank %d of %d differs from the single-device model" % (r, len(bounds) - 1)


# ---- real multi-process RCCL (needs two GPUs: skipped on the one-GPU test box, run by whoever has the 8-GPU node) ------------
def _rccl_rank(rank, world, uid, q, fail_rank, timeout_s):
    """One process per GPU.  Trains target 4 row-sharded; `fail_rank` leaves before its first collective (a dead peer)."""
    import os
    import sys
    os.environ["RGBM_COMM_TIMEOUT_S"] = str(timeout_s)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "spark-data-repair-plugin_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from repair import _native as N
    from tests.synth import make_table, balanced_weights
    try:
        dirty, clean, cards = make_table(60000, 8, seed=71, null_ratio=0.02)
        feats = [c for c in range(8) if c != 4]
        K = int(cards[4])
        kw = dict(objective=1, num_class=K, n_estimators=6, learning_rate=0.2, device_id=rank)
        N.comm_init(uid, rank, world, rank)
        if rank == fail_rank:
            q.put((rank, "left", None))
            return                                       # never joins the collectives: the peers must not hang
        b0, b1 = rank * 60000 // world, (rank + 1) * 60000 // world
        tab = N.Table(np.ascontiguousarray(dirty[:, b0:b1]), cards, device_id=rank)
        blob = tab.train(4, feats, class_weight=balanced_weights(dirty[4], K), row_sharded=True, **kw).save()
        single = None
        if rank == 0:
            N.comm_finalize()
            single = N.Table(dirty, cards, device_id=0).train(4, feats, class_weight=balanced_weights(dirty[4], K), **kw).save()
        q.put((rank, "ok", (blob, single)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", "%s: %s" % (type(e).__name__, e)))


def _run_ranks(world, fail_rank=-1, timeout_s=120):
    import multiprocessing as mp
    from repair import _native as N
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = N.comm_unique_id()
    ps = [ctx.Process(target=_rccl_rank, args=(r, world, uid, q, fail_rank, timeout_s)) for r in range(world)]
    for p in ps:
        p.start()
    out = {}
    for _ in range(world):
        r, status, payload = q.get(timeout=600)
        out[r] = (status, payload)
    for p in ps:
        p.join(timeout=60)
    return out


def _two_gpus():
    from repair import _native as N
    return N.device_count() >= 2


def test_rccl_two_processes_give_the_single_device_model(level_pass_kind):
    if not _two_gpus():
        pytest.skip("needs two HIP devices (one process per GPU over RCCL)")
    out = _run_ranks(2)
    assert all(s == "ok" for s, _ in out.values()), out
    single = out[0][1][1]
    assert out[0][1][0] == single and out[1][1][0] == single


def test_rccl_dead_peer_raises_instead_of_hanging(level_pass_kind):
    """A rank that never enters the collectives: the surviving rank's watchdog (RGBM_COMM_TIMEOUT_S) aborts its communicator and
    the training call fails with an error instead of blocking in ncclAllReduce for ever."""
    if not _two_gpus():
        pytest.skip("needs two HIP devices (one process per GPU over RCCL)")
    out = _run_ranks(2, fail_rank=1, timeout_s=8)
    assert out[1][0] == "left"
    assert out[0][0] == "error" and ("collective" in out[0][1] or "RCCL" in out[0][1]), out


def _rccl_fused_rank(rank, world, uid, q):
    """One process per GPU: a fusion group of two members (targets 4 and 0) over the real RCCL communicator."""
    import os
    import sys
    import threading
    os.environ["RGBM_COMM_TIMEOUT_S"] = "120"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "spark-data-repair-plugin_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from repair import _native as N
    from tests.synth import make_table, balanced_weights
    try:
        dirty, clean, cards = make_table(60000, 8, seed=71, null_ratio=0.02)
        N.comm_init(uid, rank, world, rank)
        b0, b1 = rank * 60000 // world, (rank + 1) * 60000 // world
        tab = N.Table(np.ascontiguousarray(dirty[:, b0:b1]), cards, device_id=rank)
        fg = N.FusionGroup(2)
        out, err = {}, []

        def member(j, t):
            try:
                with fg.member(j):
                    feats = [c for c in range(8) if c != t]
                    K = int(cards[t])
                    out[t] = tab.train(t, feats, class_weight=balanced_weights(dirty[t], K), row_sharded=True, objective=0 if K == 2 else 1, num_class=max(K, 2),
                                       n_estimators=6, learning_rate=0.2, device_id=rank).save()
            except Exception as e:  # noqa: BLE001
                err.append("%s: %s" % (type(e).__name__, e))
        ths = [threading.Thread(target=member, args=(j, t)) for j, t in enumerate((4, 0))]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=300)
        info = fg.info()
        fg.close()
        N.comm_finalize()
        single = {}
        if rank == 0 and not err:
            full = N.Table(dirty, cards, device_id=0)
            for t in (4, 0):
                feats = [c for c in range(8) if c != t]
                K = int(cards[t])
                single[t] = full.train(t, feats, class_weight=balanced_weights(dirty[t], K), objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=6,
                                       learning_rate=0.2, device_id=0).save()
        q.put((rank, "error" if err else "ok", (err or out, single, info)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", ("%s: %s" % (type(e).__name__, e), {}, {})))


def test_rccl_two_processes_fusion_group(level_pass_kind):
    """The fusion group (two row-sharded targets in flight per rank, one all-reduce per step) over REAL RCCL ranks: both models on both ranks are the
    single-device models.  Needs two GPUs, like the tests above."""
    if not _two_gpus():
        pytest.skip("needs two HIP devices (one process per GPU over RCCL)")
    import multiprocessing as mp
    from repair import _native as N
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = N.comm_unique_id()
    ps = [ctx.Process(target=_rccl_fused_rank, args=(r, 2, uid, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in range(2):
        r, status, payload = q.get(timeout=600)
        res[r] = (status, payload)
    for p in ps:
        p.join(timeout=60)
    assert all(s == "ok" for s, _ in res.values()), res
    single = res[0][1][1]
    for r in (0, 1):
        for t in (4, 0):
            assert res[r][1][0][t] == single[t], "rank %d target c%d" % (r, t)
    assert res[0][1][2]["parts"] == res[1][1][2]["parts"] > res[0][1][2]["collectives"]
