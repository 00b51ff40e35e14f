"""cProfile of pipeline.search_on_table (48 fits, batch of 16 evaluations) on the HIP engine: where the host time goes."""
import cProfile, pstats, os, sys, io
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from tests.synth import make_table
from repair.engine import HipEngine
from repair.pipeline import search_on_table, encode_frame
dirty, clean, cards = make_table(10000, 8, seed=5, null_ratio=0.0)
X = pd.DataFrame({"c%d" % c: ["v%d" % v for v in clean[c]] for c in range(8) if c != 4})
df = X.assign(y=pd.Series(["k%d" % v for v in clean[4]]))
idx, remaps, dicts = encode_frame(df, list(df.columns))
eng = HipEngine(0)
opts = {"model.hp.max_evals": "16", "model.hp.no_progress_loss": "100", "model.hp.batch_size": "16"}
base = dict(n_estimators=300, learning_rate=0.01, max_depth=7, max_bin=255, seed=42)
for rep in range(2):
    tab = eng.upload_dictionaries(idx, remaps)
    pr = cProfile.Profile(); pr.enable()
    search_on_table(eng, tab, len(df.columns) - 1, np.asarray(tab.n_codes), base, opts)
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
