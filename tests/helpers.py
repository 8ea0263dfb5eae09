"""Seeded synthetic inputs shared by the tests (numpy side = oracle inputs)."""
import numpy as np


def random_csr(rng, n_rows, n_cols, avg_deg, hubs=(), empty_rows=(), weighted=True):
    """Random bipartite CSR [n_rows, n_cols]; ``hubs`` = {row: degree}; sorted columns."""
    deg = rng.poisson(avg_deg, size=n_rows).astype(np.int64)
    deg = np.minimum(deg, n_cols)
    for r, d in dict(hubs).items():
        deg[r] = min(d, n_cols)
    for r in empty_rows:
        deg[r] = 0
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.empty(rowptr[-1], dtype=np.int64)
    for i in range(n_rows):
        d = deg[i]
        if d:
            if d * 4 > n_cols:
                col[rowptr[i]:rowptr[i + 1]] = np.sort(rng.permutation(n_cols)[:d])
            else:
                c = np.unique(rng.integers(0, n_cols, size=d * 2))[:d]
                while c.size < d:
                    c = np.unique(np.concatenate([c, rng.integers(0, n_cols, size=d)]))[:d]
                col[rowptr[i]:rowptr[i + 1]] = c
    val = rng.standard_normal(col.size).astype(np.float32) if weighted else None
    return rowptr, col, val


def tolerance(y64, atol=1e-4, rtol=1e-4):
    """Elementwise bound |y - y64| <= atol + rtol * |y64| (fp32 vs fp64 reference)."""
    return atol + rtol * np.abs(y64)
