"""-m gpu: the MULTI-RANK job flow of engine.run_job on the HIP engine -- what `bench.py --config 100m32 --gpus N` runs -- on the one GPU the
test box has.

A rank of the shard-only job holds nothing but its row shard: every target trains row-sharded over all ranks (integer all-reduces inside
librepairgbm, the rank's targets in flight together in a fusion group), the rank repairs the dirty rows of its own shard, C1 (serialised
models) and C2 (repaired cells) are all-gathered -- on the library's communicator, from device buffers, when it spans the job.  Until
round 6 only bench.py flags exercised this on hardware, and tests/test_dist_gloo.py with the oracle engine on CPU.  Here:
  * an RCCL world of one (ncclCommInitRank with one rank: real RCCL calls, ncclAllReduce / ncclAllGather included), and
  * two and three THREAD ranks (repair.dist.ThreadWorld + _native.LocalGroup: one rank per host thread on device 0)
run the same run_job() and must give the plain single-rank job's models, labels and probabilities, bit for bit.
Reference flow: python/repair/model.py:768-815 (one model per target), 1062-1143 (chained repair), 1069 (broadcast), 1142 (union).
"""
import threading

import numpy as np
import pytest

from tests.synth import make_table

pytestmark = pytest.mark.gpu

PARAMS = dict(num_leaves=31, max_depth=7, max_bin=255, min_data_in_leaf=20, min_data_in_bin=3, bagging_freq=0, seed=42, learning_rate=0.1,
              lambda_l1=0.0, lambda_l2=0.0, min_gain_to_split=0.0, min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0, n_estimators=6)


def _inputs(rows=60000, cols=10, seed=17):
    dirty, _, cards = make_table(rows, cols, seed=seed, null_ratio=0.02)
    targets = [0, 1, 4, 7, 9]                                     # binary, K = 3, 8, 24, 48
    counts = {t: np.bincount(dirty[t][dirty[t] >= 0], minlength=int(cards[t])) for t in targets}
    return dirty, cards, targets, counts


def _plain(dirty, cards, targets, counts):
    from repair.engine import HipEngine, run_job
    eng = HipEngine(0)
    mask = (dirty[targets] < 0).any(axis=0)
    res = run_job(eng, eng.upload(dirty, cards), eng.upload(np.ascontiguousarray(dirty[:, mask]), cards), cards, targets, counts, dict(PARAMS))
    return res


def _same(res, ref, targets):
    for t in targets:
        assert res["models"][t] == ref["models"][t], "model of target %d differs from the single-rank job" % t
    assert np.array_equal(res["labels"], ref["labels"]) and np.array_equal(res["probs"], ref["probs"])


def test_shard_only_job_over_an_rccl_world_of_one():
    from repair import _native as N
    from repair.engine import HipEngine, run_job
    dirty, cards, targets, counts = _inputs()
    ref = _plain(dirty, cards, targets, counts)
    eng = HipEngine(0)
    mask = (dirty[targets] < 0).any(axis=0)
    N.comm_init(N.comm_unique_id(), 0, 1, 0)
    try:
        tab = eng.upload(dirty, cards)
        res = run_job(eng, None, eng.upload(np.ascontiguousarray(dirty[:, mask]), cards), cards, targets, counts, dict(PARAMS), row_table=tab,
                      force_row_sharding=True, row_shard_all=True, dirty_is_shard=True)
        assert sorted(res["row_sharded_targets"]) == sorted(targets) and res["fusion"].get("members", 0) >= 2
        _same(res, ref, targets)
        # C1 / C2 primitives on the RCCL communicator itself (ncclAllGather with one rank)
        assert [bytes(x) for x in N.comm_all_gather_bytes(b"abc")] == [b"abc"]
        models = [N.Model.load(ref["models"][t]) for t in targets]
        feats = [[c for c in range(dirty.shape[0]) if c != t] for t in targets]
        dt = eng.upload(np.ascontiguousarray(dirty[:, mask]), cards)
        lab, prob, row0, rows = dt.repair_chain_gather(models, targets, feats)
        assert row0 == 0 and rows == [int(mask.sum())] and np.array_equal(lab, ref["labels"]) and np.array_equal(prob, ref["probs"])
        assert N.comm_gather_stats()["collectives"] >= 3
    finally:
        N.comm_finalize()


@pytest.mark.parametrize("bounds,shard_only", [([0, 25000, 60000], True), ([0, 20000, 20001, 60000], True), ([0, 31000, 60000], False)])
def test_thread_ranks_run_the_multi_rank_job_and_match_the_single_rank_job(bounds, shard_only):
    """shard_only: every rank holds its row shard only (all targets row-sharded, dirty rows of the shard, gather of unequal parts);
    else the hybrid job (whole tables on every rank, the expensive targets row-sharded, the rest target-sharded, C2 over equal shards)."""
    from repair import _native as N
    from repair import dist
    from repair.engine import HipEngine, run_job
    dirty, cards, targets, counts = _inputs()
    ref = _plain(dirty, cards, targets, counts)
    nr = len(bounds) - 1
    world, group = dist.ThreadWorld(nr), N.LocalGroup(nr)
    out, err = [None] * nr, [None] * nr
    mask = (dirty[targets] < 0).any(axis=0)

    def work(r):
        try:
            with world.rank(r):
                group.join(r)
                try:
                    eng = HipEngine(0)
                    shard = np.ascontiguousarray(dirty[:, bounds[r]:bounds[r + 1]])
                    if shard_only:
                        dm = (shard[targets] < 0).any(axis=0)
                        dtab = eng.upload(np.ascontiguousarray(shard[:, dm]), cards) if dm.any() else None       # (a 1-row shard may hold no dirty row)
                        out[r] = run_job(eng, None, dtab, cards, targets, counts, dict(PARAMS),
                                         row_table=eng.upload(shard, cards), row_shard_all=True, dirty_is_shard=True)
                    else:
                        out[r] = run_job(eng, eng.upload(dirty, cards), eng.upload(np.ascontiguousarray(dirty[:, mask]), cards), cards, targets, counts, dict(PARAMS),
                                         row_table=eng.upload(shard, cards))
                    out[r]["gather_via"] = dist.GATHER["via"]
                finally:
                    N.comm_finalize()
        except Exception as e:  # noqa: BLE001
            err[r] = e

    ths = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=900)
    real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    for e in real + [e for e in err if e is not None]:       # (the rank that failed first, not the peers its failure released from a barrier)
        raise e
    first = 0
    for r in range(nr):
        _same(out[r], ref, targets)
        assert out[r]["gather_via"].startswith("librepairgbm"), out[r]["gather_via"]      # C1 / C2 rode on the library's communicator
        if shard_only:
            assert out[r]["dirty_row0"] == first
            first += int((dirty[targets][:, bounds[r]:bounds[r + 1]] < 0).any(axis=0).sum())
        else:
            assert len(out[r]["row_sharded_targets"]) >= 1 and len(out[r]["row_sharded_targets"]) < len(targets)
