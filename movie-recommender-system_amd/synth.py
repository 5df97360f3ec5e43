"""Synthetic inputs of the benchmark configurations (no dataset files exist).

``interactions``  implicit-feedback matrix: lognormal user degree (sigma 1,
                  given mean) x Zipf(0.9) item popularity over a random
                  permutation of the items, distinct items per user, value 1.0
                  -- the shape of the reference's data: binary fp32 CSR
                  (``als/als.py:118-122``) at the scale its README states
                  (~1 M users x ~27 k movies).
``lightfm_problem``  LightFM-shaped inputs: identity + 3 categorical tokens per
                  user (7 ages, 21 jobs, 2 genders: the vocabulary at
                  ``hybrid_docker/inference.py:168-177``), identity + ~6.7
                  Zipf-distributed tokens per item, nc = 15 embeddings.

Degrees and popularity are drawn with numpy (PCG64) on the host; the bulk
sampling and de-duplication run as torch ops on whatever device is given, in
chunks of 65 536 users with a random stream each, so the 5e7-entry case takes
about a second on the GPU and any row range can be generated on its own.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import scipy.sparse as sp
import torch


GEN_CHUNK = 1 << 16   # users per generation chunk: each chunk has a random stream of its own


def _popularity_cdf(rng, n_items: int) -> np.ndarray:
    rank = rng.permutation(n_items)
    pop = (rank + 10.0) ** -0.9
    pop /= pop.sum()
    return np.cumsum(pop)


def interactions_torch(seed: int, n_users: int, n_items: int, mean_degree: float, device="cpu",
                       users: Optional[Tuple[int, int]] = None):
    """Returns (indptr int64 [rows+1], indices int32 [nnz], data fp32 [nnz]) as
    torch tensors on `device`; rows sorted by column, no duplicates.

    ``users = (u0, u1)`` returns only those rows of the same matrix (``indptr``
    then starts at 0): popularity and degrees are drawn on the host for all
    users, the item draws in chunks of ``GEN_CHUNK`` users with one random
    stream per chunk, so that a rank of a sharded run -- or a caller that walks
    a large matrix slab by slab -- generates its rows and nothing else, and gets
    the rows every other split gets."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cdf_h = _popularity_cdf(rng, n_items)
    mu = np.log(mean_degree) - 0.5
    deg = np.clip(np.rint(rng.lognormal(mu, 1.0, size=n_users)), 1, max(1, n_items // 4)).astype(np.int64)
    u0, u1 = (0, n_users) if users is None else (int(users[0]), int(users[1]))
    assert 0 <= u0 <= u1 <= n_users
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    cdf = torch.from_numpy(cdf_h).to(dev)
    cdf[-1] = 1.0
    cols_parts, count_parts = [], []
    for ch in range(u0 // GEN_CHUNK, -(-u1 // GEN_CHUNK) if u1 > u0 else u0 // GEN_CHUNK):
        c0, c1 = ch * GEN_CHUNK, min((ch + 1) * GEN_CHUNK, n_users)
        # chunk 0 keeps the plain seed: matrices of up to GEN_CHUNK users are those of the unchunked generator
        gen.manual_seed(int(seed) if ch == 0 else (int(seed) * 1_000_003 + ch) % (1 << 62))
        deg_t = torch.from_numpy(deg[c0:c1]).to(dev)
        local = torch.repeat_interleave(torch.arange(c1 - c0, device=dev, dtype=torch.int64), deg_t)
        draws = torch.rand(local.numel(), device=dev, dtype=torch.float64, generator=gen)
        items = torch.searchsorted(cdf, draws).clamp_(max=n_items - 1)
        keys = torch.unique(local * n_items + items)  # sorted, duplicates (repeat draws) dropped
        rows = torch.div(keys, n_items, rounding_mode="floor")
        cols = (keys - rows * n_items).to(torch.int32)
        a, b = max(u0, c0) - c0, min(u1, c1) - c0      # the chunk's rows that were asked for
        if (a, b) != (0, c1 - c0):
            keep = (rows >= a) & (rows < b)
            rows, cols = rows[keep], cols[keep]
        count_parts.append(torch.bincount(rows - a, minlength=b - a))
        cols_parts.append(cols)
        del local, draws, items, keys, rows
    indptr = torch.zeros(u1 - u0 + 1, dtype=torch.int64, device=dev)
    if count_parts:
        indptr[1:] = torch.cumsum(torch.cat(count_parts), 0)
    cols = torch.cat(cols_parts) if cols_parts else torch.zeros(0, dtype=torch.int32, device=dev)
    data = torch.ones(cols.numel(), dtype=torch.float32, device=dev)
    return indptr, cols, data


def interactions(seed: int, n_users: int, n_items: int, mean_degree: float) -> sp.csr_matrix:
    """Host (scipy) form of `interactions_torch`, for tests and small cases."""
    indptr, cols, data = interactions_torch(seed, n_users, n_items, mean_degree, "cpu")
    return sp.csr_matrix((data.numpy(), cols.numpy(), indptr.numpy()), shape=(n_users, n_items))


def split_holdout(matrix: sp.csr_matrix, seed: int, fraction: float = 0.2) -> Tuple[sp.csr_matrix, sp.csr_matrix]:
    """Random per-entry train / held-out split (both CSR, same shape)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    coo = matrix.tocoo()
    held = rng.random(coo.nnz) < fraction
    mk = lambda m: sp.csr_matrix((coo.data[m], (coo.row[m], coo.col[m])), shape=matrix.shape, dtype=np.float32)
    return mk(~held), mk(held)


def lightfm_problem(seed: int, n_users: int, n_items: int, no_components: int = 15, item_vocab: int = 6744,
                    item_tokens_mean: float = 6.7):
    """Returns a dict with user_features [U, U+30], item_features [I, I+vocab]
    (identity block first, as lightfm.data.Dataset lays them out) and random
    embeddings N(0, 1/nc), biases N(0, 0.1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sizes = (7, 21, 2)
    n_uf = n_users + sum(sizes)
    rows = np.repeat(np.arange(n_users), 4)
    cols = np.empty((n_users, 4), dtype=np.int64)
    cols[:, 0] = np.arange(n_users)
    off = n_users
    for j, s in enumerate(sizes):
        cols[:, j + 1] = off + rng.integers(0, s, size=n_users)
        off += s
    user_features = sp.csr_matrix((np.ones(n_users * 4, dtype=np.float32), (rows, cols.ravel())),
                                  shape=(n_users, n_uf))
    ntok = rng.poisson(item_tokens_mean, size=n_items) + 1
    tok_p = (np.arange(item_vocab) + 1.0) ** -1.0
    tok_p /= tok_p.sum()
    tok = rng.choice(item_vocab, size=int(ntok.sum()), p=tok_p)
    irow = np.concatenate([np.arange(n_items), np.repeat(np.arange(n_items), ntok)])
    icol = np.concatenate([np.arange(n_items), n_items + tok])
    item_features = sp.csr_matrix((np.ones(len(irow), dtype=np.float32), (irow, icol)),
                                  shape=(n_items, n_items + item_vocab))
    item_features.sum_duplicates()
    item_features.data[:] = 1.0
    nc = no_components
    return {
        "user_features": user_features, "item_features": sp.csr_matrix(item_features),
        "user_embeddings": (rng.standard_normal((n_uf, nc)) / np.sqrt(nc)).astype(np.float32),
        "user_biases": (rng.standard_normal(n_uf) * 0.1).astype(np.float32),
        "item_embeddings": (rng.standard_normal((n_items + item_vocab, nc)) / np.sqrt(nc)).astype(np.float32),
        "item_biases": (rng.standard_normal(n_items + item_vocab) * 0.1).astype(np.float32),
    }
