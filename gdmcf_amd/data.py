"""Synthetic interaction matrices of the Yelp / Amazon-Book / stress shapes.

The reference ships no dataset, so the benchmark uses the nominal published shapes of
the DiffRec "clean" splits with this generator: user degree ~ clipped lognormal (sigma 1.0) with
the mean chosen to hit the target nnz, items drawn without replacement per user with popularity
p_i ~ (i+1)^-0.8, numpy Generator(PCG64(seed)).  CSR int32 indices, sorted columns.
"""
import numpy as np

SHAPES = {
    "yelp": dict(n_users=54574, n_items=34395, nnz=981915),
    "amazon-book": dict(n_users=108822, n_items=94949, nnz=2202379),
    "stress": dict(n_users=1000000, n_items=200000, nnz=20000000),
    "tiny": dict(n_users=6000, n_items=3001, nnz=90000),  # launcher / data-parallel rehearsals in tests (ragged width)
}


def synth_degrees(n_users, n_items, nnz, rng):
    mean = nnz / n_users
    mu = np.log(mean) - 0.5  # lognormal mean = exp(mu + sigma^2/2), sigma = 1
    d = np.clip(np.round(rng.lognormal(mu, 1.0, n_users)), 4, n_items // 4).astype(np.int64)
    return d


def synth_user_rows(n_items, degrees, rng):
    """Column ids per user: weighted sampling WITHOUT replacement from p_i ~ (i+1)^-0.8, done as
    successive sampling (i.i.d. inverse-CDF draws, repeats skipped, first d distinct kept)."""
    p = np.arange(1, n_items + 1, dtype=np.float64) ** -0.8
    cdf = np.cumsum(p / p.sum())
    cdf[-1] = 1.0
    out = []
    for d in degrees:
        d = int(d)
        got = np.empty(0, dtype=np.int64)
        while got.size < d:
            draw = np.searchsorted(cdf, rng.random(2 * d + 16), side="right")
            cat = np.concatenate([got, draw])
            _, first = np.unique(cat, return_index=True)
            got = cat[np.sort(first)]
        out.append(np.sort(got[:d]).astype(np.int32))
    return out


def synth_csr(shape="yelp", n_rows=None, seed=0):
    """CSR (indptr int64, indices int32) for the first `n_rows` users of the named shape."""
    cfg = SHAPES[shape]
    rng = np.random.Generator(np.random.PCG64(seed))
    n_users = cfg["n_users"] if n_rows is None else min(n_rows, cfg["n_users"])
    deg = synth_degrees(cfg["n_users"], cfg["n_items"], cfg["nnz"], rng)[:n_users]
    rows = synth_user_rows(cfg["n_items"], deg, rng)
    indptr = np.zeros(n_users + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return indptr, np.concatenate(rows).astype(np.int32), cfg["n_items"]


def dense_batches(indptr, indices, n_items, batch_size, n_batches):
    """float32 {0,1} dense batches [n_batches, B, I] (the layout the reference's DataLoader yields)."""
    out = np.zeros((n_batches, batch_size, n_items), dtype=np.float32)
    for b in range(n_batches):
        for r in range(batch_size):
            u = b * batch_size + r
            out[b, r, indices[indptr[u]:indptr[u + 1]]] = 1.0
    return out
