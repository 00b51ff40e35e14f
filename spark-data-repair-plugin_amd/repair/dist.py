"""Multi-GPU sharding of the repair hot path: one process per GPU, torch.distributed (backend
"nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What the reference does with Spark (python/repair/model.py):
  * model-parallel training, one task per target attribute (`_build_repair_stat_models_in_parallel`,
    model.py:817-926, models come back pickled, :910,:921)            -> ``assign_targets`` (LPT)
  * `sparkContext.broadcast(models)` (model.py:1069)                  -> ``exchange_blobs`` (all-gather
    of the serialised models: a few MB, latency-bound, no all-reduce anywhere on this path)
  * data-parallel inference over random row groups (model.py:1054-1058,1142)
                                                                      -> ``shard_rows`` + ``gather_rows``
The chain semantics of `_repair` (a later model reads cells repaired by an earlier one) stay intact
because every rank holds ALL models and runs the whole chain on its own rows.

Beyond the reference: a multiclass target is one indivisible unit for target sharding (a K=64 target is
27 % of the synthetic workload, which caps 8 GPUs at 3.75x).  ``split_targets`` therefore sends the
expensive targets to ROW-sharded training over all ranks (every rank trains the same model on its row
shard; librepairgbm all-reduces integer histograms over RCCL, see include/rgbm.h) and keeps target
sharding for the cheap ones.
"""
import threading

import numpy as np

_tls = threading.local()


class ThreadWorld:
    """The ranks of a job as host THREADS of one process: `with world.rank(r): run_job(...)` on thread r.  Every function of this module
    then sees a world of `n` ranks, and the exchanges run through this object (or, for the data path, through the thread's librepairgbm
    communicator -- _native.LocalGroup -- when it spans the world).  Exists so that the multi-rank job logic (engine.run_job: target / row
    sharding, fusion group, C1 / C2) can be driven on the ONE GPU the test box has; torch.distributed is process-global and cannot."""

    def __init__(self, n):
        self.n = int(n)
        self._bar = threading.Barrier(self.n)
        self._slots = [None] * self.n

    class _Rank:
        def __init__(self, w, r):
            self.w, self.r = w, r

        def __enter__(self):
            _tls.world = (self.w, self.r)
            return self

        def __exit__(self, *exc):
            _tls.world = None
            if exc and exc[0] is not None:
                self.w._bar.abort()          # a failing rank must not leave its peers in a barrier
            return False

    def rank(self, r):
        return ThreadWorld._Rank(self, int(r))

    def all_gather(self, r, obj):
        self._slots[r] = obj
        self._bar.wait(timeout=600)
        out = list(self._slots)
        self._bar.wait(timeout=600)
        return out


def _thread_world():
    return getattr(_tls, "world", None)


def _lib_comm_spans_world():
    """The calling thread's librepairgbm communicator (RCCL, or the thread group of the tests) covers exactly the ranks of the job: the data
    path of C1 / C2 then runs on it -- device buffers, ncclAllGather over xGMI -- and torch's process group only carries control traffic."""
    rank, ws = world()
    if ws <= 1:
        return False
    try:
        from repair import _native
        ci = _native.comm_info()
    except Exception:  # noqa: BLE001 - no library (CPU tests with the oracle engine)
        return False
    return ci["kind"] in (1, 2) and ci["nranks"] == ws and ci["rank"] == rank


GATHER = {"via": "none", "bytes": 0, "seconds": 0.0, "collectives": 0}     # what bench.py prints as config.rccl.gather


def _note_gather(via):
    GATHER["via"] = via
    if via.startswith("librepairgbm"):
        try:
            from repair import _native
            st = _native.comm_gather_stats()
            GATHER.update(bytes=st["bytes"], seconds=st["seconds"], collectives=st["collectives"])
        except Exception:  # noqa: BLE001
            pass


def _all_gather_bytes(payload):
    """Variable-size all-gather of one uint8 array per rank -> list in rank order.  The one primitive under C1 (model blobs) and C2 (repaired
    cells): on the library's communicator when it spans the world, else through the thread world / torch's process group."""
    payload = np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
    tw = _thread_world()
    if _lib_comm_spans_world():
        from repair import _native
        out = _native.comm_all_gather_bytes(payload)
        _note_gather("librepairgbm communicator (device buffers, all-gather)")
        return out
    if tw is not None:
        return tw[0].all_gather(tw[1], payload.copy())
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return [payload]
    import torch
    dev = _tensor_device()
    ws = d.get_world_size()
    n = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    d.all_gather(ns, n)
    counts = [int(x.item()) for x in ns]
    mx = max(max(counts), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if payload.size:
        buf[:payload.size] = torch.from_numpy(payload.copy()).to(dev)
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(ws)]
    d.all_gather(bufs, buf)
    _note_gather("torch.distributed (%s)" % d.get_backend())
    return [bufs[r].cpu().numpy()[:counts[r]].copy() for r in range(ws)]


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:  # pragma: no cover - torch always importable here
        pass
    return None


def world():
    tw = _thread_world()
    if tw is not None:
        return (tw[1], tw[0].n)
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def assign_targets(costs, world_size):
    """Longest-processing-time-first assignment of target attributes to ranks.

    costs: list of (target, cost) -- cost ~ trees per iteration * training rows.
    Returns a list (per rank) of target lists; deterministic (ties by original order)."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i][1]), i))
    loads = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (loads[j], j))
        out[r].append(costs[i][0])
        loads[r] += float(costs[i][1])
    return out


# What one boosting iteration of ONE target costs however few rows a rank holds, in the plan's cost units (class trees x 10^6 training
# rows): the level grower is a chain of ~35 dependent kernels (plan / split-find / reduce / replay around the passes) of ~1.5 ms per
# iteration on MI355X (binary target, 10M rows: 1.9-3.0 ms per iteration measured, profiles/r04e_*), where a K = 64 target on 10M rows
# (640 units) takes 27 ms: 1.5 ms ~ 36 units; a row-sharded target adds one integer all-reduce per level (~7 x 30 us ~ 5 units).
LAUNCH_FLOOR_UNITS = 36.0
COLLECTIVE_UNITS = 5.0


def split_targets(costs, world_size, row_sharding, force=False, all_targets=False):
    """(big, small): targets whose cost exceeds a QUARTER of a rank's fair share are trained row-sharded over all ranks (in the
    given order, identical on every rank); the rest is LPT-assigned rank by rank.

    A row-sharded target costs every rank 1 / world_size of its units and nothing else (the per-level all-reduce is <= 8 MB), a
    target-sharded one sits on ONE rank whole.  With the round-2 threshold (half a fair share) the 100M x 32 job on 8 ranks kept
    the K in {2, 3, 4} targets whole: 66 / 8 + 4 = 12.25 of 74 units on the critical path = 6.04x before any overhead.  At a
    quarter only the binary target stays whole: 73 / 8 + 1 = 10.1 units = 7.3x (`plan` prints this for any job).
    force: apply the rule of a 2-rank job even with one rank (single-GPU dry run of the collective path).
    all_targets: every target is row-sharded (a job whose ranks hold nothing but their row shard)."""
    if not row_sharding or (world_size <= 1 and not force):
        return [], list(costs)
    if all_targets:
        return list(costs), []
    total = float(sum(c for _, c in costs))
    thr = total / (4.0 * max(world_size, 2))
    big = [(t, c) for t, c in costs if float(c) > thr]
    small = [(t, c) for t, c in costs if not float(c) > thr]
    return big, small


def plan(costs, world_size, row_sharding, force=False, all_targets=False, fused_targets=6):
    """The schedule `engine.run_job` will follow, in cost units (class trees x training rows), so that it can be checked without
    hardware: which targets are row-sharded, the load every rank gets from the target-sharded ones, the critical path and the
    speed-up it allows before collective / tail costs -- and, when the costs are given in class trees x 10^6 rows, the same with the
    per-iteration launch floor of every target and the per-level collectives of the row-sharded ones charged (`*_with_floor`): a
    12.5 M-row shard of a 3-class target is launch-bound, and eight ranks do not make it eight times faster."""
    big, small = split_targets(costs, world_size, row_sharding, force=force, all_targets=all_targets)
    total = float(sum(c for _, c in costs))
    assign = assign_targets(small, world_size)
    cost_of = dict(costs)
    loads = [float(sum(cost_of[t] for t in a)) for a in assign]
    shared = float(sum(c for _, c in big)) / max(world_size, 1)
    path = shared + (max(loads) if loads else 0.0)
    # with the launch floor.  The row-sharded targets of a rank train in a FUSION GROUP (engine.run_job): up to `fused` of them in flight at
    # once, their chains of small kernels overlapping, and the i-th collective of all of them is ONE all-reduce -- the launch floor and the
    # collectives are charged once per round of `fused` targets (the largest target's floor), not once per target
    ws = max(world_size, 1)
    floor_of = lambda c: max(float(c), LAUNCH_FLOOR_UNITS)   # noqa: E731
    fused = max(1, int(fused_targets))
    shared_f = 0.0
    for i in range(0, len(big), fused):
        grp = [float(c) / ws for _, c in big[i:i + fused]]
        shared_f += max(sum(grp), LAUNCH_FLOOR_UNITS) + (COLLECTIVE_UNITS if ws > 1 else 0.0)
    def in_rounds(cs):      # `fused` targets in flight at once on a rank (six by default, engine._train_concurrency): the floor once per round
        cs = sorted((float(c) for c in cs), reverse=True)
        return float(sum(max(sum(cs[i:i + fused]), LAUNCH_FLOOR_UNITS) for i in range(0, len(cs), fused)))
    loads_f = [in_rounds(cost_of[t] for t in a) for a in assign]
    path_f = shared_f + (max(loads_f) if loads_f else 0.0)
    single_f = in_rounds(c for _, c in costs)
    del floor_of
    return dict(world_size=world_size, total_units=total, row_sharded=[t for t, _ in big], row_sharded_units_per_rank=shared,
                target_sharded=assign, target_sharded_units_per_rank=loads, critical_path_units=path,
                ideal_speedup=(total / path) if path > 0 else 1.0,
                critical_path_units_with_floor=path_f, speedup_with_floor=(single_f / path_f) if path_f > 0 else 1.0,
                fused_row_sharded_targets=fused, collectives_per_iteration=(8 * ((len(big) + fused - 1) // fused)) if ws > 1 else 0)


ROW_COMM = {"asked": False, "ranks": 1, "init_sec": 0.0, "fell_back": False, "why": ""}     # what bench.py prints as config.rccl


def init_row_comm(device_id):
    """Create the librepairgbm RCCL communicator of this process (one rank per GPU).  Returns True when every rank
    succeeded AND the communicator spans the whole world (ncclCommCount); on any failure every rank tears down and the job
    falls back to plain target sharding.  ROW_COMM records what happened (ranks seen, seconds, fell back and why)."""
    import time
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return False
    import torch
    from repair import _native
    rank, ws = d.get_rank(), d.get_world_size()
    ROW_COMM.update(asked=True, ranks=1, fell_back=False, why="")
    t0 = time.perf_counter()
    box = [_native.comm_unique_id() if rank == 0 else None]
    d.broadcast_object_list(box, src=0)
    ok, why = 1, ""
    try:
        _native.comm_init(box[0], rank, ws, device_id)
        seen = _native.comm_count()
        ROW_COMM["ranks"] = seen
        if seen != ws:
            ok, why = 0, "the communicator spans %d ranks, the job has %d" % (seen, ws)
    except Exception as e:  # noqa: BLE001 - any failure means "no row sharding", never a crash
        ok, why = 0, "%s: %s" % (type(e).__name__, e)
    t = torch.tensor([ok], dtype=torch.int32, device=_tensor_device())
    d.all_reduce(t, op=d.ReduceOp.MIN)
    ROW_COMM["init_sec"] = time.perf_counter() - t0
    if int(t.item()) == 1:
        return True
    ROW_COMM.update(fell_back=True, why=why or "another rank could not create its communicator")
    if _native.comm_info()["kind"] != 0:
        _native.comm_finalize()
    return False


def shard_rows(n_rows, world_size, rank):
    """Contiguous, balanced row shard [begin, begin+count) of rank."""
    base, rem = divmod(int(n_rows), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def _tensor_device():
    import torch
    d = _dist()
    if d is not None and d.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def exchange_blobs(local):
    """All-gather a dict {key(int) -> bytes} so that every rank ends up with the union (C1: the serialised models)."""
    if world()[1] == 1:
        return dict(local)
    keys = sorted(local)
    # header: number of blobs, then (key, length) pairs; then the blobs
    head = np.array([len(keys)] + [v for k in keys for v in (k, len(local[k]))], np.int64)
    payload = np.concatenate([np.array([head.size], np.int64).view(np.uint8), head.view(np.uint8)] + [np.frombuffer(local[k], np.uint8) for k in keys])
    out = {}
    for raw in _all_gather_bytes(payload):
        nh = int(raw[:8].view(np.int64)[0])
        h = raw[8:8 + nh * 8].view(np.int64)
        off = 8 + nh * 8
        for i in range(int(h[0])):
            k, ln = int(h[1 + 2 * i]), int(h[2 + 2 * i])
            out[k] = raw[off:off + ln].tobytes()
            off += ln
    return out


def gather_rows(local, n_rows_total):
    """All-gather row shards: local [T][rows_of_rank] -> [T][n_rows_total] (shard order = rank order)."""
    if world()[1] == 1:
        return np.asarray(local)
    res, _ = gather_rows_var(local)
    assert res.shape[1] == n_rows_total, (res.shape, n_rows_total)
    return res


def gather_rows_var(local):
    """All-gather row shards whose sizes only the owning rank knows (every rank repairs the dirty rows of ITS row shard):
    local [T][rows_of_rank] -> ([T][sum of the ranks' rows] in rank order, first row of this rank)."""
    local = np.ascontiguousarray(local)
    rank, ws = world()
    if ws == 1:
        return local, 0
    T = local.shape[0]
    parts = [p.view(local.dtype).reshape(T, -1) if p.size else np.zeros((T, 0), local.dtype) for p in _all_gather_bytes(local)]
    return np.concatenate(parts, axis=1), int(sum(p.shape[1] for p in parts[:rank]))


def sum_arrays(a):
    """Element-wise sum of an integer / float array over the ranks (label counts of row shards)."""
    a = np.ascontiguousarray(a)
    tw = _thread_world()
    if tw is not None:
        return np.sum(np.stack(tw[0].all_gather(tw[1], a)), axis=0).astype(a.dtype)
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return a
    import torch
    t = torch.from_numpy(a.astype(np.float64 if a.dtype.kind == "f" else np.int64)).to(_tensor_device())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return t.cpu().numpy().astype(a.dtype)


def barrier():
    tw = _thread_world()
    if tw is not None:
        tw[0].all_gather(tw[1], None)
        return
    d = _dist()
    if d is not None:
        d.barrier()


def max_over_ranks(x):
    tw = _thread_world()
    if tw is not None:
        return float(max(tw[0].all_gather(tw[1], float(x))))
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=_tensor_device())
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(x):
    """The smallest value any rank holds: how ranks settle on a number derived from rank-local inputs (engine.run_job: fusion members)."""
    return -max_over_ranks(-float(x))


def sum_over_ranks(x):
    tw = _thread_world()
    if tw is not None:
        return float(sum(tw[0].all_gather(tw[1], float(x))))
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=_tensor_device())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return float(t.item())
