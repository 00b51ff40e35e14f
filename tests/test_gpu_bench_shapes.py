"""-m gpu: the BENCHMARKED shapes under the parity gate.

bench.py quotes its numbers on BASELINE configs[2] (10M rows x 16 columns) and, per GPU of the 8-GPU job, on a row shard of
configs[3] (100M x 32: 12.5M rows x 32 columns -> two 16-feature chunks).  The other parity tests stop at 2.5M rows, so these
two hold the HIP trainer to the oracle at the real shapes: the K = 64 and the binary target of configs[2], and the K = 24
target of a configs[3] shard, two boosting iterations each, serialised model bytes identical.  The oracle runs its histograms
feature-parallel (OpenMP; bit-identical for any thread count), which keeps each case to a minute or two of host time.
"""
import os

import numpy as np
import pytest

from tests.synth import make_table, balanced_weights

pytestmark = pytest.mark.gpu


def _both(dirty, cards, target, iters):
    from oracle import oracle as O
    from repair import _native as N
    cols = dirty.shape[0]
    feats = [c for c in range(cols) if c != target]
    K = int(cards[target])
    cw = balanced_weights(dirty[target], K)
    kw = dict(objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters)       # everything else: the reference's defaults
    mg = N.Table(dirty, cards).train(target, feats, class_weight=cw, **kw).save()           # resident-table path, rows with a NULL target skipped on the device
    rows = dirty[target] >= 0
    O.lib().orc_set_threads(min(32, os.cpu_count() or 1))
    try:
        mo = O.train(np.ascontiguousarray(dirty[feats][:, rows]), cards[feats], dirty[target][rows], K, class_weight=cw, **kw).save()
    finally:
        O.lib().orc_set_threads(1)
    return mg, mo


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("target", [10, 0])
def test_config2_shape_10m_x_16(target):
    dirty, clean, cards = make_table(10_000_000, 16, seed=42)
    del clean
    assert int(cards[target]) == (64 if target == 10 else 2)
    mg, mo = _both(dirty, cards, target, iters=2)
    assert mg == mo, "10M x 16, target c%d: HIP model differs from the oracle" % target


@pytest.mark.timeout(1800)
def test_config3_per_gpu_shape_12_5m_x_32():
    dirty, clean, cards = make_table(12_500_000, 32, seed=43)
    del clean
    target = 7
    assert int(cards[target]) == 24
    mg, mo = _both(dirty, cards, target, iters=2)
    assert mg == mo, "12.5M x 32 (two feature chunks), target c7: HIP model differs from the oracle"
