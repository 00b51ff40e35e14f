"""Synthetic categorical tables of SURVEY.md 8(d) -- the benchmark inputs of BASELINE.json configs[2] / configs[3]:
latent-cluster dependency (z ~ U{0..63}; column c = perm_c[z] mod card_c with probability 0.9, else uniform),
cardinalities cycling through CARDS, i.i.d. NULL injection (mirrors RepairMiscApi.injectNullAt,
src/main/scala/org/apache/spark/api/python/RepairMiscApi.scala:155-182: IF(rand() > ratio, col, NULL)).
Codes are what the encoder (repair/encode.py) would produce for the values "c{c}_v{code}": 0..card-1, NULL = -1."""
import numpy as np

CARDS = [2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def make_table(n_rows, n_cols, seed, null_ratio=0.01, cards=None):
    """Returns (codes_with_nulls [C][N] int32, clean_codes [C][N] int32, n_codes [C]).  One PCG64 stream, column by column."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cards = [CARDS[c % len(CARDS)] for c in range(n_cols)] if cards is None else list(cards)
    z = rng.integers(0, 64, n_rows, dtype=np.int32)
    clean = np.empty((n_cols, n_rows), np.int32)
    for c in range(n_cols):
        perm = rng.permutation(64).astype(np.int32)
        v = perm[z] % cards[c]
        noise = rng.random(n_rows) < 0.1
        v = np.where(noise, rng.integers(0, cards[c], n_rows, dtype=np.int32), v)
        clean[c] = v
    dirty = clean.copy()
    if null_ratio > 0:
        for c in range(n_cols):
            dirty[c][rng.random(n_rows) < null_ratio] = -1
    return dirty, clean, np.asarray(cards, np.int32)


def make_table_parallel(n_rows, n_cols, seed, null_ratio=0.01, cards=None, threads=16, chunk=4_000_000, row_range=None):
    """The same distribution for the 100M-row shape: every (column, row chunk) draws from its own spawned PCG64 stream so that the
    columns can be filled by a thread pool (numpy releases the GIL inside the generators).  Deterministic in (seed, chunk), but NOT
    the stream of make_table.  Returns (dirty [C][N] int32, null_truth {col: (rows, clean codes)}, n_codes) -- the clean table is kept
    only where a cell was nulled (that is all the accuracy check needs; a second 12.8 GB array is not).
    row_range = (begin, end): only those rows of the SAME table are generated (a rank of a multi-GPU job builds its own row shard:
    the chunks that overlap the range are drawn, the others are never touched); `dirty` is then [C][end - begin] and the rows of
    null_truth are still positions in the whole table."""
    from concurrent.futures import ThreadPoolExecutor
    cards = [CARDS[c % len(CARDS)] for c in range(n_cols)] if cards is None else list(cards)
    root = np.random.SeedSequence(seed)
    zs, perm_ss, col_ss = root.spawn(3)
    nchunks = (n_rows + chunk - 1) // chunk
    rb, re_ = (0, n_rows) if row_range is None else (max(0, int(row_range[0])), min(n_rows, int(row_range[1])))
    ci0, ci1 = (rb // chunk, (max(re_, rb + 1) + chunk - 1) // chunk) if re_ > rb else (0, 0)     # chunks that overlap the range
    base = ci0 * chunk                                                                            # first row held in memory
    n_held = max(0, min(n_rows, ci1 * chunk) - base)
    z = np.empty(n_held, np.int8)
    zseeds = zs.spawn(nchunks)

    def fill_z(i):
        b, e = i * chunk, min(n_rows, (i + 1) * chunk)
        z[b - base:e - base] = np.random.Generator(np.random.PCG64(zseeds[i])).integers(0, 64, e - b, dtype=np.int8)
    perms = [np.random.Generator(np.random.PCG64(s)).permutation(64).astype(np.int32) for s in perm_ss.spawn(n_cols)]
    dirty = np.empty((n_cols, n_held), np.int32)
    cseeds = [s.spawn(nchunks) for s in col_ss.spawn(n_cols)]
    truth = {c: [] for c in range(n_cols)}

    def fill(job):
        c, i = job
        b, e = i * chunk, min(n_rows, (i + 1) * chunk)
        rng = np.random.Generator(np.random.PCG64(cseeds[c][i]))
        v = (perms[c][z[b - base:e - base]] % cards[c]).astype(np.int32)
        noise = rng.random(e - b) < 0.1
        v[noise] = rng.integers(0, cards[c], int(noise.sum()), dtype=np.int32)
        if null_ratio > 0:
            nul = np.flatnonzero(rng.random(e - b) < null_ratio)
            truth[c].append((i, nul + b, v[nul].copy()))
            v[nul] = -1
        dirty[c, b - base:e - base] = v
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(fill_z, range(ci0, ci1)))
        list(ex.map(fill, [(c, i) for c in range(n_cols) for i in range(ci0, ci1)]))
    null_truth = {}
    for c in range(n_cols):
        parts = sorted(truth[c], key=lambda p: p[0])
        null_truth[c] = (np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, np.int64),
                         np.concatenate([p[2] for p in parts]) if parts else np.zeros(0, np.int32))
    if row_range is not None:
        dirty = np.ascontiguousarray(dirty[:, rb - base:re_ - base])
        for c in range(n_cols):
            r, v = null_truth[c]
            keep = (r >= rb) & (r < re_)
            null_truth[c] = (r[keep], v[keep])
    return dirty, null_truth, np.asarray(cards, np.int32)


def balanced_weights(y, n_classes):
    """sklearn class_weight='balanced' (reference train.py:39-40,105) from a label column (NULL = -1 ignored)."""
    cnt = np.bincount(y[y >= 0], minlength=n_classes).astype(np.float64)
    n = cnt.sum()
    nz = (cnt > 0).sum()
    with np.errstate(divide="ignore"):
        w = np.where(cnt > 0, n / (nz * cnt), 0.0)
    return w
