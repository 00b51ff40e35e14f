"""Synthetic categorical tables of SURVEY.md 8(d): latent-cluster dependency, cycling cardinalities,
i.i.d. NULL injection (mirrors RepairMiscApi.injectNullAt: IF(rand() > ratio, col, NULL))."""
import numpy as np

CARDS = [2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64]


def make_table(n_rows, n_cols, seed, null_ratio=0.01, cards=None):
    """Returns (codes_with_nulls [C][N] int32, clean_codes [C][N] int32, n_codes [C])."""
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


def balanced_weights(y, n_classes):
    cnt = np.bincount(y[y >= 0], minlength=n_classes).astype(np.float64)
    n = cnt.sum()
    nz = (cnt > 0).sum()
    with np.errstate(divide="ignore"):
        w = np.where(cnt > 0, n / (nz * cnt), 0.0)
    return w
