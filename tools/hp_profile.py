import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
import numpy as np, pandas as pd
from tests.synth import make_table
from repair.train import build_model
dirty, clean, cards = make_table(10000, 8, seed=5, null_ratio=0.0)
X = pd.DataFrame({"c%d" % c: ["v%d" % v for v in clean[c]] for c in range(8) if c != 4})
y = pd.Series(["k%d" % v for v in clean[4]])
opts = {"model.hp.max_evals": "6", "model.hp.no_progress_loss": "100", "model.hp.batch_size": "1"}
build_model(X, y, True, int(cards[4]), n_jobs=-1, opts=opts)
pr = cProfile.Profile(); pr.enable()
build_model(X, y, True, int(cards[4]), n_jobs=-1, opts=opts)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
