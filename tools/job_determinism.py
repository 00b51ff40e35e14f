"""The bench job (engine.run_job: 16 target models, six in flight, then the chained repair) twice in one process and once more after
other allocations: models and labels must be identical every time."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from repair.engine import HipEngine, run_job
from repair.synth import make_table
from tests.numerics_bound import first_differing_iteration
rows, iters, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
BASE = dict(num_leaves=31, max_depth=7, max_bin=255, min_data_in_leaf=20, min_data_in_bin=3, bagging_freq=0, seed=42, learning_rate=0.01, lambda_l1=0.0, lambda_l2=0.0,
            min_gain_to_split=0.0, min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0, n_estimators=iters)
dirty, clean, cards = make_table(rows, 16, seed=42)
targets = list(range(16))
mask = (dirty[targets] < 0).any(axis=0)
dirty_rows = np.ascontiguousarray(dirty[:, mask])
eng = HipEngine(0)
tab = eng.upload(dirty, cards)
counts = {t: np.bincount(dirty[t][dirty[t] >= 0], minlength=int(cards[t])) for t in targets}
ref = None
for rep in range(reps):
    if rep == reps - 1:   # perturb the allocator state: a few unrelated buffers of odd sizes
        junk = [eng.upload(np.ascontiguousarray(dirty[:, :100003 + 7919 * i]), cards) for i in range(5)]
        del junk
    fresh = eng.upload(dirty_rows, cards)
    if len(sys.argv) > 4 and rep >= int(sys.argv[4]):
        os.environ["RGBM_PREDICTOR"] = "walk"          # the remaining repeats score with the index-linked walk instead of the bit-vector tables
    t0 = time.time()
    res = run_job(eng, tab, fresh, cards, targets, counts, BASE)
    out = (res["models"], res["labels"].copy(), res["probs"].copy())
    if ref is None:
        ref = out
        print("repeat 0: %.1fs" % (time.time() - t0), flush=True)
        continue
    badm = [(t, first_differing_iteration(ref[0][t], out[0][t])[0]) for t in targets if ref[0][t] != out[0][t]]
    dl = np.argwhere(ref[1] != out[1]); dp = np.argwhere(ref[2] != out[2])
    print("repeat %d: %.1fs, differing models %s, differing labels %d, differing probabilities %d %s" % (rep, time.time() - t0, badm or "none", len(dl), len(dp), dp[:3].tolist()), flush=True)
