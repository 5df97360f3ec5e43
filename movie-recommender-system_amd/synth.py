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
sampling and de-duplication run as torch ops on whatever device is given, so
the 5e7-entry case takes about a second on the GPU.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np
import scipy.sparse as sp
import torch


def interactions_torch(seed: int, n_users: int, n_items: int, mean_degree: float, device="cpu"):
    """Returns (indptr int64 [U+1], indices int32 [nnz], data fp32 [nnz]) as
    torch tensors on `device`; rows sorted by column, no duplicates."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rank = rng.permutation(n_items)
    pop = (rank + 10.0) ** -0.9
    pop /= pop.sum()
    mu = np.log(mean_degree) - 0.5
    deg = np.clip(np.rint(rng.lognormal(mu, 1.0, size=n_users)), 1, max(1, n_items // 4)).astype(np.int64)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    cdf = torch.from_numpy(np.cumsum(pop)).to(dev)
    cdf[-1] = 1.0
    deg_t = torch.from_numpy(deg).to(dev)
    users = torch.repeat_interleave(torch.arange(n_users, device=dev, dtype=torch.int64), deg_t)
    draws = torch.rand(users.numel(), device=dev, dtype=torch.float64, generator=gen)
    items = torch.searchsorted(cdf, draws).clamp_(max=n_items - 1)
    keys = torch.unique(users * n_items + items)  # sorted, duplicates (repeat draws) dropped
    rows = torch.div(keys, n_items, rounding_mode="floor")
    cols = (keys - rows * n_items).to(torch.int32)
    counts = torch.bincount(rows, minlength=n_users)
    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(counts, 0)
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
