"""Where does `RepairModel.run()` spend its time on a synthetic frame?  (< 10 % in pandas.)

    python tools/resident_probe.py [--rows 1000000] [--cols 8] [--estimators 300] [--engine hip|oracle] [--profile]

Prints the wall-clock of run(), the device-pipeline phases recorded by the resident path and the remaining host share."""
import argparse, cProfile, os, pstats, sys, time
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spark-data-repair-plugin_amd"))
from repair.errors import NullErrorDetector
from repair.model import RepairModel
from repair.synth import make_table

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000); ap.add_argument("--cols", type=int, default=8)
ap.add_argument("--estimators", type=int, default=300); ap.add_argument("--engine", default="hip"); ap.add_argument("--profile", action="store_true")
ap.add_argument("--categorical", action="store_true", help="hand the columns over as pandas Categorical (dictionary-encoded) instead of Python string objects")
a = ap.parse_args()
dirty, clean, cards = make_table(a.rows, a.cols, seed=13, null_ratio=0.01)
df = pd.DataFrame({"tid": np.arange(a.rows)})
for c in range(a.cols):
    v = np.array(["c%d_v%02d" % (c, k) for k in range(int(cards[c]))], object)[np.maximum(dirty[c], 0)]
    v[dirty[c] < 0] = None
    df["c%d" % c] = pd.Categorical(v) if a.categorical else v
m = RepairModel().setInput(df).setRowId("tid").setErrorDetectors([NullErrorDetector()])
for k, v in {"model.hp.max_evals": "1", "model.lgb.n_estimators": str(a.estimators), "model.max_training_row_num": str(a.rows)}.items():
    m = m.option(k, v)
if a.engine == "oracle":
    from tests.helpers import OracleEngine
    m._engine_override = OracleEngine()
prof = cProfile.Profile() if a.profile else None
for rep in range(2):   # the second run is the warm one (library load, allocator, kernel code)
    t0 = time.perf_counter()
    if prof and rep == 1: prof.enable()
    out = m.run()
    if prof and rep == 1: prof.disable()
    wall = time.perf_counter() - t0
    info = m._last_resident_info
    assert info is not None, "run() did not take the resident path"
    dev = {k: round(v, 3) for k, v in info["times"].items()}
    device = sum(info["times"].get(k, 0.0) for k in ("train", "infer", "detect", "prepare", "exchange", "gather"))
    print("run %d (%s columns): %d rows x %d cols, %d repaired cells: %.2f s wall; device pipeline %.2f s %s; host share %.1f %%"
          % (rep, "categorical" if a.categorical else "object", a.rows, a.cols, len(out), wall, device, dev, 100 * (1 - device / wall)), flush=True)
if prof:
    pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
