"""hp-search latency at the reference's default sample size (10 000 rows): sequential vs batched over HIP streams."""
import os, sys, time
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table
from repair.train import build_model
dirty, clean, cards = make_table(10000, 8, seed=5, null_ratio=0.0)
X = pd.DataFrame({"c%d" % c: ["v%d" % v for v in clean[c]] for c in range(8) if c != 4})
y = pd.Series(["k%d" % v for v in clean[4]])
base = {"model.hp.max_evals": "16", "model.hp.no_progress_loss": "100"}
res = {}
for bs in (() if os.environ.get("HP_PROBE_RESIDENT_ONLY") else ("1", "4", "8", "16")):
    t0 = time.perf_counter()
    (m, score), _ = build_model(X, y, True, int(cards[4]), n_jobs=-1, opts=dict(base, **{"model.hp.batch_size": bs}))
    dt = time.perf_counter() - t0
    res[bs] = (score, m.booster_bytes_)
    print("batch_size=%s: 16 evaluations x 3 folds + final fit, 300 iterations each: %.2f s  (cv f1=%.4f)" % (bs, dt, score), flush=True)
if res:
    assert len(set(res.values())) == 1, "batched search changed the outcome"
    print("identical outcome for every batch size")

# ---- the same search on RESIDENT tables (pipeline.search_on_table): folds are device row gathers, no re-encode / upload per fit
from repair.engine import HipEngine
from repair.pipeline import search_on_table, encode_frame
eng = HipEngine(0)
df = X.assign(y=y)
idx, remaps, dicts = encode_frame(df, list(df.columns))
for bs in ("1", "8", "16"):
    tab = eng.upload_dictionaries(idx, remaps)
    t0 = time.perf_counter()
    best = search_on_table(eng, tab, len(df.columns) - 1, np.asarray(tab.n_codes), dict(n_estimators=300, learning_rate=0.01, max_depth=7, max_bin=255, seed=42),
                           dict(base, **{"model.hp.batch_size": bs}))
    dt = time.perf_counter() - t0
    print("resident tables, batch_size=%s: 16 evaluations x 3 folds (48 fits, 300 iterations each): %.2f s  best=%s" % (bs, dt, best), flush=True)
