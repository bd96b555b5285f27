"""Seeded synthetic URMs of the shapes BASELINE.json / SURVEY.md section 8(d) name (no dataset can be downloaded here).

Generator family: per-user activity ~ lognormal clipped to [min_len, max_len] and rescaled to the target nnz;
item popularity p_i ~ (i+1)^-0.8 (Zipf-like); duplicates removed per user; every item gets at least one
interaction (no cold rows/columns, see SURVEY section 7 hard part 7).  `binary` values are all 1.0; `real` values are
ratings U{1..5} plus 1e-3*U(0,1) jitter so that top-K comparisons are tie-free.
"""
import numpy as np
import scipy.sparse as sps

SHAPES = {
    #  name      n_users  n_items  nnz          min_len max_len  seed
    "ml1m":    (6040,    3706,    1000209,     20,     2314,    20190916),
    "ml20m":   (138493,  26744,   20000263,    20,     9254,    20190917),
    "netflix": (480189,  17770,   100480507,   1,      17653,   20190918),
}


def synthetic_urm(n_users, n_items, nnz, min_len=1, max_len=None, seed=0, values="binary", zipf_exponent=0.8):
    rng = np.random.default_rng(seed)
    max_len = min(max_len or n_items - 1, n_items - 1)
    raw = rng.lognormal(mean=0.0, sigma=1.0, size=n_users)
    lens = raw / raw.sum() * nnz
    lens = np.clip(np.rint(lens), min_len, max_len).astype(np.int64)
    # a second rescale pass brings the total close to the target after clipping
    free = (lens > min_len) & (lens < max_len)
    if free.any():
        deficit = nnz - lens.sum()
        lens[free] = np.clip(np.rint(lens[free] * (1.0 + deficit / max(1, lens[free].sum()))), min_len, max_len).astype(np.int64)
    p = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), zipf_exponent)
    cdf = np.cumsum(p / p.sum())
    total = int(lens.sum())
    # oversample 1.5x per user, then de-duplicate per row and trim to the wanted length
    over = np.minimum((lens * 3) // 2 + 8, n_items)
    rows = np.repeat(np.arange(n_users, dtype=np.int64), over)
    cols = np.searchsorted(cdf, rng.random(int(over.sum())), side="right").astype(np.int64)
    cols = np.minimum(cols, n_items - 1)
    key = np.unique(rows * n_items + cols)             # sorted by (row, col), duplicates removed
    rows, cols = key // n_items, key % n_items
    # keep at most lens[u] entries of each row, chosen at random among the unique ones
    order = np.lexsort((rng.random(len(rows)), rows))
    rows, cols = rows[order], cols[order]
    first = np.searchsorted(rows, np.arange(n_users))
    rank = np.arange(len(rows)) - first[rows]
    keep = rank < lens[rows]
    rows, cols = rows[keep], cols[keep]
    # no cold items: give every missing item one interaction from a random user
    missing = np.setdiff1d(np.arange(n_items), np.unique(cols))
    if len(missing):
        rows = np.concatenate([rows, rng.integers(0, n_users, len(missing))])
        cols = np.concatenate([cols, missing])
    if values == "binary":
        data = np.ones(len(rows), dtype=np.float32)
    else:
        data = (rng.integers(1, 6, len(rows)) + 1e-3 * rng.random(len(rows))).astype(np.float32)
    urm = sps.csr_matrix((data, (rows, cols)), shape=(n_users, n_items), dtype=np.float32)
    urm.sum_duplicates()
    if values == "binary":
        urm.data[:] = 1.0
    urm.sort_indices()
    del total
    return urm


def named_urm(name, values="binary", scale=1.0):
    """`scale` < 1 shrinks users, items and nnz proportionally (smaller parity-test cases of the same family)."""
    n_users, n_items, nnz, min_len, max_len, seed = SHAPES[name]
    if scale != 1.0:
        n_users = max(8, int(n_users * scale)); n_items = max(8, int(n_items * scale))
        nnz = max(n_users, int(nnz * scale * scale)); max_len = max(min_len + 1, int(max_len * scale))
        min_len = max(1, min(min_len, n_items // 4))
    return synthetic_urm(n_users, n_items, nnz, min_len, max_len, seed, values)
