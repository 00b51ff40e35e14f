#!/usr/bin/env python3
"""bench.py -- repaired cells/sec of the repair-model hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 10m16|100m32]

Workloads (config.workload):
  10m16  (default) BASELINE.json configs[2]: synthetic 10M rows x 16 categorical columns, 1 % injected NULLs, seed 42, every
         column a target attribute (what NullErrorDetector yields).  The largest config quoted for ONE GPU.
  100m32 BASELINE.json configs[3], the north-star multi-GPU shape: 100M rows x 32 columns, seed 43, target attributes c0..c7
         (31 feature columns each).  Fits one GPU (~60 GB), so `--gpus 1 --config 100m32` is the base of its scaling curve.
Both use the reference's fixed LightGBM parameters (train.py:102-115) + LightGBM defaults for the searched ones and train on
ALL rows (model.max_training_row_num = N).

Timing contract: W untimed warm-up boosting iterations, then EXACTLY K boosting iterations ("steps") of EVERY target model
(+ the chained repair of every dirty row + the result exchange) between barrier + synchronize brackets, max over ranks ->
`ms_per_step`.  The reference's job is n_estimators = 300 iterations (train.py:53-55), so the headline `value` is always the
cells/sec of the COMPLETE 300-iteration job: when K != 300 that job is run (and timed the same way) right after the K-step
region; `value_basis` says which run the number comes from.  --no-full-job skips it (value then describes the K-step job and
says so).  With --gpus N (N > 1) and no torchrun environment the script launches its N ranks itself.

Multi-GPU ("scaling": "strong", the SAME job on more GPUs): expensive targets are trained row-sharded over all ranks (every
rank holds a row shard, librepairgbm all-reduces integer histograms over RCCL -- the model is bit-identical for any N), cheap
ones are target-sharded (LPT), the chained repair is row-sharded and the repaired cells are all-gathered from device buffers
on the same RCCL communicator (one per rank; torch's process group is gloo and carries control traffic only).
--mode targets = pure target sharding (the reference's own parallel mode).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

BASE_PARAMS = dict(num_leaves=31, max_depth=7, max_bin=255, min_data_in_leaf=20, min_data_in_bin=3,
                   bagging_freq=0, seed=42, learning_rate=0.01, lambda_l1=0.0, lambda_l2=0.0,
                   min_gain_to_split=0.0, min_sum_hessian_in_leaf=1e-3, bagging_fraction=1.0, feature_fraction=1.0)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
# The LDS-atomic floor of the histogram build ("the ceiling"): a 64-bit LDS atomic instruction of one wave costs ~12.8 cycles of its
# CU's LDS unit when its lanes do not pile up on addresses (measured on the root pass and by tools/lds_atomic_lanes.hip: profiles/r01_lds_atomic_microbench.txt,
# r03d_*; the unit's own floor is 10.3), the chip has 256 CUs at 2.4 GHz.  A launch that accumulates P rows with A atomics each cannot finish before
# P * A / 64 * 12.8 / (256 * 2.4e9) seconds, whatever its HBM traffic is.
LDS_ATOMIC_CYCLES, N_CUS, CLOCK_HZ = 12.8, 256, 2.4e9
REF_N_ESTIMATORS = 300         # model.lgb.n_estimators default, python/repair/train.py:53-55
CONFIGS = {
    "10m16": dict(rows=10_000_000, cols=16, seed=42, n_targets=16, baseline="configs[2]"),
    "100m32": dict(rows=100_000_000, cols=32, seed=43, n_targets=8, baseline="configs[3]"),
}


def cpu_baseline(cols, seed, targets, steps_full, budget_s=25.0, full_rows=0, full_iters=20, cells_full=0):
    """CPU legs on a bounded sample of the same workload, timed on this box's host cores.

    "port": oracle/rgbm_oracle.c, the plain-C restatement of the LightGBM 3.3.1 path (the real Spark + LightGBM stack cannot be
    installed here: BASELINE.md section 2).  One target attribute per host thread (the reference's per-target parallel mode) and
    LightGBM-style feature-parallel histograms inside every fit (OpenMP), so that all cores are busy.
    "hgb": scikit-learn's HistGradientBoostingClassifier (an independent LightGBM-family implementation, all cores through
    its own OpenMP pool) on the same sample -- BASELINE.md's B2."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from repair.engine import balanced_class_weight
    from repair.synth import make_table
    n, iters = 250_000, 10
    dirty, clean, cards = make_table(n, cols, seed=seed)
    nproc = os.cpu_count() or 1
    par = max(1, min(len(targets), nproc))
    per_fit = max(1, nproc // par)
    O.lib().orc_set_threads(per_fit)
    feats_of = {t: [c for c in range(cols) if c != t] for t in targets}

    def fit(t):
        r = dirty[t] >= 0
        K = int(cards[t])
        cw = balanced_class_weight(np.bincount(dirty[t][r], minlength=K))
        return O.train(np.ascontiguousarray(dirty[feats_of[t]][:, r]), cards[feats_of[t]], dirty[t][r], K, class_weight=cw,
                       objective=0 if K == 2 else 1, num_class=max(K, 2), n_estimators=iters, **BASE_PARAMS)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=par) as ex:
        models = list(ex.map(fit, targets))
    t_train = time.perf_counter() - t0
    O.lib().orc_set_threads(1)
    mask = (dirty[targets] < 0).any(axis=0)
    dr = np.ascontiguousarray(dirty[:, mask])
    cells = int((dr[targets] < 0).sum())
    t0 = time.perf_counter()
    O.repair_chain(models, list(targets), [feats_of[t] for t in targets], [list(range(int(cards[t]))) for t in targets], dr)
    t_infer = time.perf_counter() - t0
    scale = steps_full / float(iters)
    out = dict(value=cells / max((t_train + t_infer) * scale, 1e-9), unit="repaired cells/sec", cores=min(nproc, par * per_fit), kind="port",
               nproc=nproc,
               sample="%d-row sample x %d cols, %d targets: %d fits in parallel x %d OpenMP threads each (feature-parallel histograms), "
                      "%d of %d boosting iterations timed (train %.2fs + single-threaded chained repair %.2fs), time scaled x%.1f; "
                      "cells/s of the sample stands for the full table only as far as cost is linear in rows (histogram build and repair are; "
                      "the per-node split search is not and is amortised better on the full table)"
                      % (n, cols, len(targets), par, per_fit, iters, steps_full, t_train, t_infer, scale))
    # ---- B2: sklearn HGB on the same sample, all cores, as many targets as the time budget allows (cost-weighted scale-up)
    try:
        from sklearn.ensemble import HistGradientBoostingClassifier
        cost = {t: (1 if cards[t] <= 2 else int(cards[t])) for t in targets}
        done, t_fit, t_pred = [], 0.0, 0.0
        for t in sorted(targets, key=lambda t: cost[t]):
            if done and t_fit + t_pred > budget_s:
                break
            r = dirty[t] >= 0
            X = dirty[feats_of[t]].T.astype(np.float64); X[X < 0] = np.nan
            t0 = time.perf_counter()
            h = HistGradientBoostingClassifier(max_iter=iters, learning_rate=0.01, max_depth=7, max_leaf_nodes=31, max_bins=255,
                                               class_weight="balanced", early_stopping=False).fit(X[r], dirty[t][r])
            t_fit += time.perf_counter() - t0
            t0 = time.perf_counter()
            h.predict(X[~r])
            t_pred += time.perf_counter() - t0
            done.append(t)
        share = sum(cost[t] for t in done) / float(sum(cost.values()))
        out["hgb"] = dict(value=cells / max((t_fit + t_pred) * scale / share, 1e-9), unit="repaired cells/sec", cores=nproc, kind="sklearn-hgb",
                          sample="same sample; targets %s fitted one after another on all cores (%.2fs fit + %.2fs predict for %d of %d iterations), "
                                 "scaled by iterations x%.1f and by their %.0f %% share of the class trees" % (done, t_fit, t_pred, iters, steps_full, scale, 100 * share))
    except Exception as e:  # noqa: BLE001 - the secondary leg never fails the bench
        out["hgb"] = dict(value=None, error=str(e))
    # ---- two targets at FULL size, unscaled rows: the binary target and a K = 8 one on the whole table, binning + `full_iters`
    # boosting iterations each, every core the feature-parallel histograms can use; from their per-iteration times an estimate of the whole job
    # whose only scaling is over class trees and iterations (cost per (row, class tree, iteration) is what both fits measure)
    try:
        if full_rows and full_rows * cols <= 400_000_000:
            fd, _, fcards = make_table(full_rows, cols, seed=seed)
            want = [targets[0]] + [t for t in targets if int(fcards[t]) >= 8][:1]
            fits, thr = [], min(nproc, 64)
            O.lib().orc_set_threads(thr)
            for t in want:
                r = fd[t] >= 0
                K = int(fcards[t])
                cw = balanced_class_weight(np.bincount(fd[t][r], minlength=K))
                Xf = np.ascontiguousarray(fd[feats_of[t]][:, r]); yf = fd[t][r]
                kw = dict(class_weight=cw, objective=0 if K == 2 else 1, num_class=max(K, 2), **BASE_PARAMS)
                t0 = time.perf_counter()
                O.train(Xf, fcards[feats_of[t]], yf, K, n_estimators=1, **kw)                 # binning + one iteration: the set-up share
                t_one = time.perf_counter() - t0
                t0 = time.perf_counter()
                O.train(Xf, fcards[feats_of[t]], yf, K, n_estimators=full_iters, **kw)
                dt = time.perf_counter() - t0
                per_it = max(dt - t_one, 1e-9) / max(full_iters - 1, 1)
                fits.append(dict(target="c%d (K=%d)" % (t, K), class_trees=1 if K == 2 else K, rows=int(r.sum()), iterations=full_iters, train_sec=dt,
                                 setup_and_first_iteration_sec=t_one, sec_per_iteration=per_it))
                del Xf, yf
            del fd
            O.lib().orc_set_threads(1)
            out["full_size_target"] = dict(fits[0], threads=thr, note="binning + %d boosting iterations of ONE target model on the whole table, not scaled" % full_iters)
            out["full_size_targets"] = fits
            if len(fits) > 1:
                trees_job = sum(1 if int(cards[t]) <= 2 else int(cards[t]) for t in targets)
                per_tree_it = fits[-1]["sec_per_iteration"] / fits[-1]["class_trees"]          # the multiclass fit: a class tree's iteration on all rows
                est = trees_job * per_tree_it * steps_full + sum(f["setup_and_first_iteration_sec"] for f in fits) / len(fits) * len(targets)
                out["full_size_estimate"] = dict(value=cells_full / est if cells_full else None, unit="repaired cells/sec", job_sec=est, threads=thr,
                                                 how="whole-table fits of %s, one after another on %d threads; job = %d class trees x %d iterations x %.3f s per class-tree iteration "
                                                     "+ set-up per target; rows NOT scaled -- the figure the GPU/CPU ratio should be read against (the 250 000-row sample above "
                                                     "amortises the split search worse and keeps the histograms in cache)" % (", ".join(f["target"] for f in fits), thr, trees_job, steps_full, per_tree_it))
    except Exception as e:  # noqa: BLE001
        out["full_size_target"] = dict(error=str(e))
    out["read_the_gpu_cpu_ratio_against"] = "full_size_estimate (whole-table fits, rows not scaled)" if "full_size_estimate" in out else "value (250 000-row sample, scaled)"
    return out


def _spawn_ranks(a):
    """--gpus N without a torchrun environment: launch the N ranks (one per GPU) exactly as the driver would."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def _load_traffic(config, world, rows, cols, train_rows, forced_collectives):
    """Measured HBM bytes per launch of the two histogram kernel classes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
    calibrated on a kernel of known traffic as MI355X_MICROARCH.md prescribes): the counters cannot be read from inside the process,
    so the numbers come from the committed profile of the same workload over ALL its target models (tools/make_traffic_json.py;
    profiles/traffic.json says which run) -- the same set of launches the algorithmic bytes next to them describe.  ONLY for the run
    the profile was taken on: the whole table of the named config on one rank, every model trained on all rows (a
    --train-rows / --rows line used to carry the whole-table counters and printed 25-60 TB/s of "HBM traffic")."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(config)
        same_run = (world == 1 and not forced_collectives and not (0 < train_rows < rows) and
                    rows == e.get("rows", CONFIGS[config]["rows"]) and cols == e.get("cols", CONFIGS[config]["cols"]))
        if e and "classes" in e and same_run:
            return e
    except Exception:  # noqa: BLE001
        pass
    return None


def true_reference_baseline(cols, seed, targets, steps_full):
    """The REAL reference arithmetic, when it can be imported: lightgbm (pinned 3.3.1 in bin/requirements.txt:6; python/repair/train.py:92,
    121-131) with the reference's fixed parameters on the bounded sample cpu_baseline() uses.  Neither this container nor the GPU box has
    the wheel (no network), so today this reports why it was skipped; tests/test_true_reference_optional.py is the parity side of the hook."""
    try:
        import lightgbm as lgb
    except Exception as e:  # noqa: BLE001
        return dict(value=None, skipped="lightgbm not importable (%s); the hook runs the day the wheel is at hand" % type(e).__name__)
    from repair.synth import make_table
    n, iters = 250_000, 10
    dirty, _, cards = make_table(n, cols, seed=seed)
    cells = int((dirty[targets] < 0).sum())
    t_all = 0.0
    for t in targets:
        feats = [c for c in range(cols) if c != t]
        r = dirty[t] >= 0
        X = dirty[feats].T.astype(np.float64); X[X < 0] = np.nan
        K = int(cards[t])
        m = lgb.LGBMClassifier(boosting_type="gbdt", objective="binary" if K <= 2 else "multiclass", class_weight="balanced", learning_rate=0.01, max_depth=7,
                               max_bin=255, reg_alpha=0.0, min_split_gain=0.0, n_estimators=iters, random_state=42, n_jobs=-1)
        t0 = time.perf_counter()
        m.fit(X[r], dirty[t][r]); m.predict(X[~r])
        t_all += time.perf_counter() - t0
    return dict(value=cells / max(t_all * steps_full / float(iters), 1e-9), unit="repaired cells/sec", cores=os.cpu_count() or 1, kind="reference",
                version=getattr(lgb, "__version__", "?"), sample="%d-row sample, %d of %d iterations, scaled" % (n, iters, steps_full))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=REF_N_ESTIMATORS, help="timed boosting iterations per target model (the reference's job is 300)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up boosting iterations")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="10m16")
    ap.add_argument("--rows", type=int, default=0, help="override the config's row count (testing)")
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-job", action="store_true", help="do not run the 300-iteration job when --steps differs from 300")
    ap.add_argument("--mode", choices=["auto", "targets"], default="auto", help="multi-GPU split: auto = hybrid row/target sharding")
    ap.add_argument("--force-row-sharding", action="store_true", help="testing: run the collective (RCCL) path with a world of one")
    ap.add_argument("--row-shard-all", action="store_true", help="testing, with --force-row-sharding: EVERY target takes the collective path (what a rank of the "
                                                                  "shard-only multi-GPU job does; with --rows 12500000 --config 100m32 this is one rank's share of the 8-GPU job)")
    ap.add_argument("--roofline-steps", type=int, default=20, help="boosting iterations of the sequential pass that measures the kernel roofline")
    ap.add_argument("--concurrency", type=int, default=None, help="target models trained at once per rank (default: RGBM_TARGET_CONCURRENCY or 6)")
    ap.add_argument("--train-rows", type=int, default=0,
                    help="train every model on a seeded sample of this many rows (the reference's DEFAULT behaviour is "
                         "model.max_training_row_num = 10000, model.py:755-766); 0 = all rows, which is what BASELINE's metric is quoted on")
    ap.add_argument("--dedup", action="store_true",
                    help="VARIANT of the workload (never the headline): train on the DISTINCT rows of the table with integer multiplicities "
                         "(repair.pipeline.distinct_rows + rgbm_table_set_row_multiplicity): byte-identical models -- models_md5 must equal the row-for-row "
                         "line's -- at the cost of the distinct rows; one rank, whole tables")
    ap.add_argument("--dump-labels", default="", help="debug: np.save the repaired labels / probabilities of the job here")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(_spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s); start it as `python bench.py --gpus N` or with "
                         "torch.distributed.run --nproc-per-node N ... bench.py --gpus N" % (a.gpus, world))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # Testing aid for boxes with fewer GPUs than ranks (BENCH_SHARE_GPU=1): the ranks share the visible devices, the process group runs
    # over gloo and the job is target-sharded (RCCL refuses two ranks on one device) -- everything but the collectives is the real path.
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = local_rank % max(1, torch.cuda.device_count())
        a.mode = "targets"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("RGBM_COMM_TIMEOUT_S", "180")   # a peer that died inside a collective fails this rank after 3 min, not 10
        import datetime
        # every wait of the job is bounded: torch's collectives by this timeout, librepairgbm's by RGBM_COMM_TIMEOUT_S (its watchdog aborts the
        # communicator and the call raises) -- a rank that is gone ends the job with an error line, not with a hang
        # The process group carries CONTROL traffic only (barriers, a few scalars, the ncclUniqueId): gloo.  Everything that moves data between
        # GPUs -- the integer all-reduces of row-sharded training, C1 (serialised models) and C2 (repaired cells) -- runs on librepairgbm's own
        # RCCL communicator (repair.dist.init_row_comm), so a rank holds ONE RCCL communicator; if that communicator cannot be created the job
        # falls back to target sharding and the gathers go through this group.
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("BENCH_DIST_TIMEOUT_S", "900"))))

    from repair import dist as rdist
    from repair.engine import HipEngine, run_job, model_params, balanced_class_weight
    from repair.synth import make_table, make_table_parallel

    cfg = dict(CONFIGS[a.config])
    rows = a.rows or cfg["rows"]
    cols = a.cols or cfg["cols"]
    targets = list(range(min(cfg["n_targets"], cols)))

    # ---- multi-GPU, large table: every rank builds, uploads and keeps ONLY ITS ROW SHARD (rows [b0, b0 + c0) of the same table: the
    # generator draws the chunks that overlap the range), every target is trained row-sharded over all ranks (integer all-reduce of
    # histograms: same models as on one GPU), the rank repairs the dirty rows of its shard and the repaired cells are all-gathered.
    # No rank ever holds the 12.8 GB table.  (Decided collectively: it needs the RCCL communicator on every rank.)
    shard_only = False
    if world > 1 and a.mode == "auto" and not share_gpu and rows * cols > 400_000_000 and not (0 < a.train_rows < rows):
        shard_only = bool(rdist.init_row_comm(local_rank))
    # ---- inputs (outside the timed region: detector + encoder outputs, resident in HBM) + the warm-up fits
    eng = HipEngine(device_id=local_rank)

    def load_and_warm(shard_only, allow_row_comm=True):
        """Generates / uploads this rank's tables and runs the warm-up fits.  Returns None when a shard-only job's row-sharded warm-up failed
        (the caller then loads whole tables on every rank and runs target-sharded: slower to set up, but the job completes)."""
        row_base = 0
        t_gen = time.perf_counter()
        if shard_only:
            row_base, c0 = rdist.shard_rows(rows, world, rank)
            dirty, null_truth, cards = make_table_parallel(rows, cols, seed=cfg["seed"], threads=max(1, min(32, os.cpu_count() or 1) // max(1, min(world, 8))),
                                                           row_range=(row_base, row_base + c0))
        elif rows * cols > 400_000_000:
            dirty, null_truth, cards = make_table_parallel(rows, cols, seed=cfg["seed"], threads=min(32, os.cpu_count() or 1) // max(1, min(world, 4)) or 1)
        else:
            dirty, clean, cards = make_table(rows, cols, seed=cfg["seed"])
            null_truth = {t: (np.flatnonzero(dirty[t] < 0), clean[t][dirty[t] < 0]) for t in targets}
            del clean
        t_gen = time.perf_counter() - t_gen
        dirty_mask = (dirty[targets] < 0).any(axis=0)
        dirty_pos = np.flatnonzero(dirty_mask) + row_base                    # positions in the whole table
        dirty_rows = np.ascontiguousarray(dirty[:, dirty_mask])
        n_cells = int((dirty_rows[targets] < 0).sum())
        n_dirty_rows = int(dirty_rows.shape[1])
        if shard_only:
            n_cells, n_dirty_rows = int(rdist.sum_over_ranks(n_cells)), int(rdist.sum_over_ranks(n_dirty_rows))
        train_src = dirty
        if 0 < a.train_rows < rows:
            sel = np.sort(np.random.Generator(np.random.PCG64(7)).choice(rows, a.train_rows, replace=False))
            train_src = np.ascontiguousarray(dirty[:, sel])
        label_counts = {t: np.bincount(train_src[t][train_src[t] >= 0], minlength=int(cards[t])) for t in targets}
        if shard_only:
            label_counts = {t: rdist.sum_arrays(label_counts[t].astype(np.int64)) for t in targets}      # GLOBAL counts (class weights, costs)
        dedup = None
        if a.dedup:
            if world > 1 or shard_only:
                raise SystemExit("bench.py --dedup: one rank, whole tables")
            from repair.pipeline import distinct_rows
            t0 = time.perf_counter()
            dist_rows, mult, _ = distinct_rows(train_src, cards)
            dedup = dict(rows=int(train_src.shape[1]), distinct_rows=int(dist_rows.shape[1]), largest_multiplicity=int(mult.max()), host_dedup_sec=time.perf_counter() - t0)
            train_src = dist_rows
        eng.upload(np.ascontiguousarray(train_src[:, :4096]), cards)   # creates the library's pinned staging ring (one-time hipHostMalloc, ~0.1 s) outside the upload figure
        torch.cuda.synchronize()
        t_up = time.perf_counter()
        train_tab = eng.upload(train_src, cards)
        if dedup is not None:
            train_tab.set_row_multiplicity(mult)
        dirty_tab = eng.upload(dirty_rows, cards)
        torch.cuda.synchronize()
        t_up = time.perf_counter() - t_up
        upload_bytes = train_src.nbytes + dirty_rows.nbytes
        row_tab = None
        if shard_only:
            row_tab = train_tab
        elif world > 1 and a.mode == "auto" and not share_gpu and allow_row_comm and rdist.init_row_comm(local_rank):
            b0, c0 = rdist.shard_rows(train_src.shape[1], world, rank)
            row_tab = eng.upload(np.ascontiguousarray(train_src[:, b0:b0 + c0]), cards)
        elif world == 1 and a.force_row_sharding:
            from repair import _native
            _native.comm_init(_native.comm_unique_id(), 0, 1, local_rank)
            row_tab = train_tab
        null_cells = dirty_rows < 0
        del dirty, train_src

        note = None
        # ---- warm-up: W boosting iterations of one model (kernel load, allocator, clocks)
        if a.warmup > 0:
            t = targets[min(4, len(targets) - 1)]
            feats = [c for c in range(cols) if c != t]
            p = dict(BASE_PARAMS, n_estimators=a.warmup)
            m = eng.train(train_tab, t, feats, balanced_class_weight(label_counts[t]), model_params(int(cards[t]), p))
            warm = eng.upload(dirty_rows[:, :min(4096, dirty_rows.shape[1])], cards)
            eng.repair_chain(warm, [m], [t], [feats], 0, warm.n)
            m_bytes = m.save()
            del warm, m
            if row_tab is not None:
                # collective warm-up (RCCL kernels, channels) that doubles as a self-check: the row-sharded model must be the
                # bytes of the single-device model trained a moment ago.  Any failure or mismatch on any rank -> every rank drops
                # the communicator and the job runs target-sharded (the reference's own parallel mode).
                ok, dig = 1, b"failed"
                try:
                    ms = eng.train_row_sharded(row_tab, t, feats, balanced_class_weight(label_counts[t]), model_params(int(cards[t]), p))
                    dig = hashlib.md5(ms.save()).digest()
                    if not shard_only:
                        ok = int(ms.save() == m_bytes)
                except Exception as e:  # noqa: BLE001
                    print("[bench] row-sharded warm-up failed on rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
                    ok = 0
                if shard_only and world > 1:   # no rank holds the whole table: the ranks must at least agree on the model, byte for byte
                    digests = rdist.exchange_blobs({rank: dig})      # (every rank takes part, also one whose fit failed: no rank waits for a missing peer)
                    ok = int(ok and len(set(digests.values())) == 1)
                if world > 1:
                    flag = torch.tensor([ok], dtype=torch.int32)
                    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                    ok = int(flag.item())
                if not ok:
                    from repair import _native
                    if _native.comm_info()["kind"] != 0:
                        _native.comm_finalize()
                    rdist.ROW_COMM.update(fell_back=True, why="the row-sharded warm-up fit failed or the ranks disagreed on its model")
                    if shard_only:      # a rank holds nothing but its shard: target sharding needs the whole table -> build it, on every rank
                        return None
                    row_tab = None
                    note = "disabled: the row-sharded warm-up model differed from the single-device one (or failed)"
        return dict(dedup=dedup, row_base=row_base, cards=cards, null_truth=null_truth, t_gen=t_gen, dirty_pos=dirty_pos, dirty_rows=dirty_rows, n_cells=n_cells,
                    n_dirty_rows=n_dirty_rows, label_counts=label_counts, train_tab=train_tab, dirty_tab=dirty_tab, t_up=t_up, upload_bytes=upload_bytes,
                    row_tab=row_tab, note=note)

    inp = load_and_warm(shard_only)
    if inp is None:
        print("[bench] rank %d: shard-only job falls back to target sharding over torch's group (every rank loads the whole table)" % rank, file=sys.stderr, flush=True)
        shard_only = False
        inp = load_and_warm(False, allow_row_comm=False)
    row_base, cards, null_truth, t_gen, dirty_pos, dirty_rows = inp["row_base"], inp["cards"], inp["null_truth"], inp["t_gen"], inp["dirty_pos"], inp["dirty_rows"]
    n_cells, n_dirty_rows, label_counts, train_tab, dirty_tab = inp["n_cells"], inp["n_dirty_rows"], inp["label_counts"], inp["train_tab"], inp["dirty_tab"]
    t_up, upload_bytes, row_tab, row_sharding_note = inp["t_up"], inp["upload_bytes"], inp["row_tab"], inp["note"]

    def plan_of(row_sharding):
        costs = [(t, (1 if int(cards[t]) <= 2 else int(cards[t])) * float(np.sum(label_counts[t])) * 1e-6) for t in targets]
        pl = rdist.plan(costs, world, row_sharding, force=a.force_row_sharding, all_targets=shard_only or (a.row_shard_all and a.force_row_sharding))
        return {k: ([round(x, 1) for x in v] if k == "target_sharded_units_per_rank" else (round(v, 2) if isinstance(v, float) else v)) for k, v in pl.items()}

    def timed_job(n_estimators, want_stats=False, concurrency=None):
        """One complete job of `n_estimators` boosting iterations per target model, bracketed as the contract says."""
        params = dict(BASE_PARAMS, n_estimators=n_estimators)
        concurrency = a.concurrency if concurrency is None else concurrency
        fresh = eng.upload(dirty_rows, cards)            # the chained repair rewrites the dirty table in place: every job starts from the NULLs
        rdist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_job(eng, train_tab, fresh, cards, targets, label_counts, params, want_stats=want_stats, row_table=row_tab,
                      force_row_sharding=a.force_row_sharding, train_concurrency=concurrency, dirty_is_shard=shard_only,
                      row_shard_all=shard_only or (a.row_shard_all and a.force_row_sharding))
        torch.cuda.synchronize(); rdist.barrier()
        return res, rdist.max_over_ranks(time.perf_counter() - t0)

    # ---- W untimed warm-up steps of the WHOLE job (a step = one boosting iteration of every target model): device-memory pool, kernel
    # code and clocks are then in the state a long-running service has them in
    if a.warmup > 0:
        timed_job(a.warmup)
    # ---- timed region: exactly K steps
    res_k, elapsed_k = timed_job(a.steps)
    # ---- the reference-configured job (n_estimators = 300), timed the same way, unless the K-step region already was that job
    run_full = a.steps != REF_N_ESTIMATORS and not a.no_full_job
    res, elapsed = timed_job(REF_N_ESTIMATORS) if run_full else (res_k, elapsed_k)
    job_steps = REF_N_ESTIMATORS if run_full else a.steps

    # ---- roofline inputs: hist_build algorithmic bytes / its summed launch time.  Every launch is bracketed by HIP events on the
    # stream it is launched on (rgbm_train_stats); with several training streams in flight a bracket also spans the other streams'
    # kernels, so the kernel figures come from a dedicated pass of the SAME job with ONE target at a time (not part of `value`).
    roof_steps = max(1, min(a.steps, a.roofline_steps))
    res_roof, elapsed_roof = timed_job(roof_steps, want_stats=True, concurrency=1)

    def agg(key, r=res_roof):
        return rdist.sum_over_ranks(sum(s.get(key, 0) for s in r["stats"]))
    hist_ms_all, hist_bytes_all, launches_all = agg("hist_ms"), agg("hist_bytes"), agg("hist_launches")
    root_ms = sum(s["root_ms"] for s in res_roof["stats"]); root_bytes = sum(s["root_rows"] * (cols - 1 + 8) for s in res_roof["stats"])
    n_root = rdist.sum_over_ranks(roof_steps * len(res_roof["stats"]))       # one root launch per boosting iteration and trained model (nchunk = 1 workloads)
    # level launches of a target x 9 B x (training-table rows of this rank) x class trees (an upper bound: finished class trees are skipped)
    level_stream_bytes = rdist.sum_over_ranks(sum((s_["hist_launches"] - roof_steps) * 9.0 * float(getattr(row_tab if shard_only else train_tab, "n", 0)) *
                                                  (1 if int(cards[s_["target"]]) <= 2 else int(cards[s_["target"]])) for s_ in res_roof["stats"]))
    train_s = rdist.max_over_ranks(res["times"]["train"]); infer_s = rdist.max_over_ranks(res["times"]["infer"])
    # what the FINISHED trees of the roofline pass needed below their roots (rows of the smaller child of every split whose children can be
    # split again): the level grower accumulates a superset (it cannot know at level 3 which nodes best-first growth will take), and
    # `roofline.achieved` counts what it accumulated -- `frac_needed` prices the same launches at the bytes LightGBM's own growth order needs
    from repair import _native as _N
    needed_level_rows = 0
    if rank == 0 and world == 1:      # (one rank: its statistics cover every launch the figure is priced over)
        needed_level_rows = sum(_N.needed_built_rows(res_roof["models"][s_["target"]], BASE_PARAMS["max_depth"]) for s_ in res_roof["stats"])

    fixed_shards = 0.0
    if shard_only:   # every rank checks the cells of its own dirty rows against the clean values it generated
        mine_l = res["labels"][:, res["dirty_row0"]:res["dirty_row0"] + dirty_rows.shape[1]]
        for i, t in enumerate(targets):
            pos, truth = null_truth[t]
            fixed_shards += float((mine_l[i][np.searchsorted(dirty_pos, pos)] == truth).sum())
        fixed_shards = rdist.sum_over_ranks(fixed_shards)
    out = None
    if rank == 0:
        labels = res["labels"]
        if a.dump_labels:
            np.save(a.dump_labels + "_labels.npy", labels); np.save(a.dump_labels + "_probs.npy", res["probs"])
            for t in targets:
                with open(a.dump_labels + "_model_%d.bin" % t, "wb") as f:
                    f.write(res["models"][t])
        fixed = int(fixed_shards)
        for i, t in enumerate(targets if not shard_only else []):
            pos, truth = null_truth[t]
            # the dirty table holds the dirty rows in ascending row order: map the nulled cells of t to their dirty-row index
            idx = np.searchsorted(dirty_pos, pos)
            fixed += int((labels[i][idx] == truth).sum())
        achieved = hist_bytes_all / max(hist_ms_all, 1e-9) * 1e-6
        traffic = _load_traffic(a.config, world, rows, cols, a.train_rows, a.force_row_sharding or a.dedup)
        # LDS atomics the launches of each class issued, target by target (the joint-bin groups of a root pass depend on the target's feature set)
        atom_ops = {"root": sum(float(s_.get("root_rows", 0)) * float(s_.get("root_atomics_per_row", 0)) for s_ in res_roof["stats"]),
                    "level": sum(float(s_.get("hist_rows", 0) - s_.get("root_rows", 0)) * float(s_.get("level_atomics_per_row", 0)) for s_ in res_roof["stats"])}
        # the two kernel classes of the histogram build, each with its own algorithmic bytes, launch time and (from the committed PMC
        # profile of the same target set) HBM-side bytes per launch
        classes = {}
        n_root_l = max(1, int(round(n_root)))
        for name, ms, nbytes, nl in (("root", root_ms, root_bytes, n_root_l), ("level", hist_ms_all - root_ms, hist_bytes_all - root_bytes, max(1, int(launches_all) - n_root_l))):
            c = {"kernel": "rg::k_level_root" if name == "root" else "rg::k_level_mt", "launches": int(nl), "avg_launch_us": ms * 1e3 / nl, "alg_bytes_per_launch": nbytes / nl,
                 "achieved": nbytes / max(ms, 1e-9) * 1e-6, "frac": nbytes / max(ms, 1e-9) * 1e-6 / HBM_PEAK_GBS}
            if name == "level":   # what a level pass streams by construction: 1 B node id + 8 B (g, h) of every (row, class tree) of its target
                c["stream_bytes_per_launch"] = level_stream_bytes / nl
                c["bound"] = "the CU's LDS pipeline: 30 ds_add_u64 per built row + ring traffic (DESIGN 5: with two steps of loads in flight per wave neither the stream nor instruction issue binds); its HBM stream alone would take stream_bytes / ~5 TB/s"
                if needed_level_rows:
                    c["needed_alg_bytes_per_launch"] = needed_level_rows * (cols - 1 + 8) / nl
                    c["frac_needed"] = needed_level_rows * (cols - 1 + 8) / max(ms, 1e-9) * 1e-6 / HBM_PEAK_GBS
            else:
                c["bound"] = "lds-atomic: 14 ds_add_u64 per row (7 joint-bin groups x (g, h)) at ~12.8 cycles per wave instruction; algorithmic bytes follow SURVEY 8(d) (F + 8 per row), the pass's HBM traffic is about half of them"
            if atom_ops[name] > 0 and world == 1:     # the LDS-atomic floor of this class: (accumulated rows x atomics per row / 64 lanes) wave instructions at the conflict-free rate
                c["atomics_per_row"] = atom_ops[name] / max(nbytes / (cols - 1 + 8), 1e-9)
                c["atomic_floor_us"] = atom_ops[name] / nl / 64.0 * LDS_ATOMIC_CYCLES / (N_CUS * CLOCK_HZ) * 1e6
                c["atomic_floor_share_of_launch"] = c["atomic_floor_us"] / max(c["avg_launch_us"], 1e-9)
            if traffic and name in traffic["classes"]:
                tc = traffic["classes"][name]
                gbps = tc["bytes_per_launch"] / max(ms / nl, 1e-9) * 1e-6
                if gbps <= HBM_PEAK_GBS:      # (a figure above the HBM peak means the profile does not describe THIS run: not evidence, not printed)
                    c["traffic"] = {"fetch_bytes_per_launch": tc["fetch_bytes_per_launch"], "write_bytes_per_launch": tc["write_bytes_per_launch"],
                                    "ratio_to_algorithmic": tc["bytes_per_launch"] / max(nbytes / nl, 1e-9), "hbm_GBps": gbps}
                else:
                    c["traffic"] = {"dropped": "profiles/traffic.json would put this class at %.0f GB/s, above the HBM peak: the committed counters are not of this run" % gbps}
                    traffic = None
            classes[name] = c
        out = {
            "metric": "repaired cells/sec", "value": n_cells / elapsed, "unit": "cells/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed_k * 1e3 / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32 gradients (LightGBM's score_t), exact int64 fixed-point histogram sums, f64 scores",
            "data": "synthetic",
            "value_basis": "complete job: n_estimators=%d per target model, %.2f s" % (job_steps, elapsed) +
                           ("" if job_steps == REF_N_ESTIMATORS else " (NOT the reference's 300-iteration job: --no-full-job)"),
            "config": {"workload": "synthetic %dM rows x %d categorical cols, 1%% NULLs, seed %d (BASELINE %s); %d target attributes, "
                                   "n_estimators=%d (reference default), %s" % (rows // 1_000_000, cols, cfg["seed"], cfg["baseline"], len(targets), REF_N_ESTIMATORS,
                                                                                 "train on all rows" if not (0 < a.train_rows < rows) else "train on a %d-row sample (reference default)" % a.train_rows),
                       "name": a.config, "rows": rows, "cols": cols, "targets": len(targets), "dirty_rows": n_dirty_rows,
                       "error_cells": n_cells,
                       "parallelism": ("row-sharded x%d: every rank generates, uploads and keeps only its row shard; all %d targets trained over all ranks (RCCL int64 "
                                       "all-reduce of histograms), dirty rows repaired where they live, repaired cells all-gathered" % (world, len(targets))) if shard_only else
                                      (("hybrid x%d: targets %s row-sharded over all ranks (RCCL int64 all-reduce of histograms), rest target-sharded"
                                        % (world, res["row_sharded_targets"])) if res["row_sharded_targets"] else "target-sharded x%d" % world),
                       # the schedule in cost units (class trees x training rows / 1e6), checkable without hardware: repair.dist.plan
                       "plan": plan_of(row_tab is not None or a.force_row_sharding)},
            "steps_region_sec": elapsed_k, "job_steps": job_steps, "elapsed_sec": elapsed,
            "model_train_sec": train_s, "repair_sec": infer_s,
            "repair_accuracy_vs_clean": fixed / max(n_cells, 1),
            # digest of every trained model of the job, in target order: two runs of the same build must print the same value
            "models_md5": hashlib.md5(b"".join(res["models"][t] for t in targets)).hexdigest(),
            "upload": {"bytes": int(upload_bytes), "sec": t_up, "GBps": upload_bytes / max(t_up, 1e-9) * 1e-9, "generate_sec": t_gen},
            "roofline": {"bound": "hbm",
                         "kernel": "rg::k_level_root + rg::k_level_mt (histogram build of the level grower; a level pass also routes the rows of its level: DataPartition::Split is not a separate kernel)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # the same launches priced at what the finished trees needed (root passes in full + the smaller child of every final split)
                         "frac_needed": ((root_bytes + needed_level_rows * (cols - 1 + 8)) / max(hist_ms_all, 1e-9) * 1e-6 / HBM_PEAK_GBS) if needed_level_rows else None,
                         "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_source": traffic["source"] if traffic else None,
                         "classes": classes,
                         "measured_on": "sequential pass (one target model at a time) of %d boosting iterations of the same job, %.2f s; HIP events per launch" % (roof_steps, elapsed_roof),
                         "launches": int(launches_all), "avg_launch_us": hist_ms_all * 1e3 / max(launches_all, 1),
                         "alg_bytes_per_launch": hist_bytes_all / max(launches_all, 1),
                         "root_scan_GBps_rank0": root_bytes / max(root_ms, 1e-9) * 1e-6,
                         # the ceiling of the two-64-bit-atomics-per-feature numerics: the same algorithmic bytes over the LDS-atomic floors of both classes
                         "atomic_floor": ({"ms_per_roofline_pass": sum(classes[n]["atomic_floor_us"] * classes[n]["launches"] for n in classes) * 1e-3,
                                           "frac_at_floor": hist_bytes_all / max(sum(classes[n]["atomic_floor_us"] * classes[n]["launches"] for n in classes) * 1e-3, 1e-9) * 1e-6 / HBM_PEAK_GBS,
                                           "cycles_per_wave_atomic": LDS_ATOMIC_CYCLES, "cus": N_CUS, "clock_hz": CLOCK_HZ,
                                           "note": "frac cannot exceed frac_at_floor with two 64-bit LDS atomics per feature and built row (DESIGN 5, the ceiling)"}
                                          if all("atomic_floor_us" in classes[n] for n in classes) else None)},
        }
        if row_sharding_note:
            out["config"]["row_sharding"] = row_sharding_note
        if inp.get("dedup"):
            # a separately named workload: same table, same models (compare models_md5 with the row-for-row line), trained on its distinct rows
            out["config"]["workload"] += "; VARIANT distinct-rows: every model trained on the %d distinct rows of the table with integer multiplicities (rgbm_table_set_row_multiplicity)" % inp["dedup"]["distinct_rows"]
            out["config"]["variant"] = dict(inp["dedup"], name="distinct-rows", note="host dedup (numpy unique of mixed-radix row keys) is outside the timed region, like encoding and upload")
            # the device counts accumulated rows in ORIGINAL rows (a row of multiplicity m counts m times): algorithmic bytes over launches that touch the
            # distinct rows only would overstate the rate -- the variant is not priced against the roofline
            out["roofline"] = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                               "note": "not priced for the distinct-rows variant: the kernels count accumulated rows in original rows; see the row-for-row line"}
        if world > 1:   # what the first multi-rank runs need to show: did RCCL really span the world, how long did it take, did the job fall back
            rc = dict(rdist.ROW_COMM)
            out["config"]["rccl"] = {"asked": rc["asked"], "ranks_seen_by_ncclCommCount": rc["ranks"], "comm_init_sec": round(rc["init_sec"], 3), "fell_back": rc["fell_back"],
                                     "why": rc["why"], "timeout_s": float(os.environ.get("RGBM_COMM_TIMEOUT_S", "600")),
                                     "fusion": res.get("fusion") or None,
                                     # C1 / C2: which path carried the model blobs and the repaired cells, and what it moved
                                     "gather": dict(rdist.GATHER)}
        if not a.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only (the other ranks would sit in the barrier below)
            out["cpu_baseline"] = cpu_baseline(cols, cfg["seed"], targets, REF_N_ESTIMATORS, full_rows=rows if not (0 < a.train_rows < rows) else 0, cells_full=n_cells)
            out["cpu_baseline"]["true_reference"] = true_reference_baseline(cols, cfg["seed"], targets, REF_N_ESTIMATORS)
            fst = out["cpu_baseline"].get("full_size_target")
            if fst and "train_sec" in fst:      # the GPU's wall time for the same target and iteration count (sequential roofline pass, setup included pro rata)
                g = [s_ for s_ in res_roof["stats"] if s_.get("target") == targets[0]]
                if g:
                    fst["gpu_total_ms_same_target_%d_iterations" % roof_steps] = g[0].get("total_ms")
    # tear the communicators down first, flush whatever the C side (RCCL prints a version banner through stdio)
    # still holds, and only then print the ONE JSON line, as the last thing this process writes
    if row_tab is not None:
        from repair import _native
        _native.comm_finalize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush(); sys.stderr.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
