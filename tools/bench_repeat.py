"""Debug aid: bench.py's exact sequence (warm-up fit, 2-iteration warm-up job, 300-iteration job) and then the 300-iteration job three more
times in the same process: model digest of every target each time."""
import hashlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
import bench
from repair.engine import HipEngine, run_job, model_params, balanced_class_weight
from repair.synth import make_table
rows = int(sys.argv[1]); iters = int(sys.argv[2]); reps = int(sys.argv[3])
dirty, clean, cards = make_table(rows, 16, seed=42)
targets = list(range(16))
mask = (dirty[targets] < 0).any(axis=0)
dirty_rows = np.ascontiguousarray(dirty[:, mask])
eng = HipEngine(0)
counts = {t: np.bincount(dirty[t][dirty[t] >= 0], minlength=int(cards[t])) for t in targets}
eng.upload(np.ascontiguousarray(dirty[:, :4096]), cards)
train_tab = eng.upload(dirty, cards)
t = 4; feats = [c for c in range(16) if c != t]
m = eng.train(train_tab, t, feats, balanced_class_weight(counts[t]), model_params(int(cards[t]), dict(bench.BASE_PARAMS, n_estimators=2)))
warm = eng.upload(dirty_rows[:, :4096], cards); eng.repair_chain(warm, [m], [t], [feats], 0, warm.n); del warm, m


def job(n):
    fresh = eng.upload(dirty_rows, cards)
    res = run_job(eng, train_tab, fresh, cards, targets, counts, dict(bench.BASE_PARAMS, n_estimators=n))
    return {t: hashlib.md5(res["models"][t]).hexdigest()[:8] for t in targets}


job(2)
ref = None
for rep in range(reps):
    d = job(iters)
    if ref is None:
        ref = d
    print("job %d: target 10 digest %s; targets differing from job 0: %s" % (rep, d[10], [t for t in targets if d[t] != ref[t]]), flush=True)
