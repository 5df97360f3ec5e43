"""float64 numpy statement of implicit-feedback ALS, written independently of
``mirec_oracle.c`` (explicit per-row normal equations instead of the
matrix-free operator) so that the two can check each other.

Test infrastructure only.  PARITY UNPINNED (see ``oracle/__init__.py``).
Follows the same call sites: ``als/als.py:166-177`` and
``src/shrekommender_system/recommenders/als_recommender.py:117-140``.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def row_system(C: sp.csr_matrix, u: int, Y: np.ndarray, YtY: np.ndarray):
    """(A_u, b_u) of row u: A = YtY + Y_u^T diag(|c|-1) Y_u, b = Y_u^T c_+."""
    s, e = C.indptr[u], C.indptr[u + 1]
    idx = C.indices[s:e]
    c = C.data[s:e].astype(np.float64)
    Yu = Y[idx].astype(np.float64)
    A = YtY + (Yu * (np.abs(c) - 1.0)[:, None]).T @ Yu
    b = Yu.T @ np.where(c > 0, c, 0.0)
    return A, b


def cg_steps_f64(A: np.ndarray, b: np.ndarray, x: np.ndarray, steps: int) -> np.ndarray:
    x = x.astype(np.float64).copy()
    r = b - A @ x
    p = r.copy()
    rsold = r @ r
    if rsold < 1e-20:
        return x
    for _ in range(steps):
        Ap = A @ p
        alpha = rsold / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        rsnew = r @ r
        if rsnew < 1e-20:
            break
        p = r + (rsnew / rsold) * p
        rsold = rsnew
    return x


def half_sweep_f64(C: sp.csr_matrix, X: np.ndarray, Y: np.ndarray, reg: float, cg_steps=3) -> np.ndarray:
    """Returns the new X (float64).  cg_steps=None solves exactly."""
    C = sp.csr_matrix(C)
    Y64 = Y.astype(np.float64)
    f = Y.shape[1]
    YtY = Y64.T @ Y64 + reg * np.eye(f)
    out = np.zeros((C.shape[0], f), dtype=np.float64)
    for u in range(C.shape[0]):
        if C.indptr[u] == C.indptr[u + 1]:
            continue
        A, b = row_system(C, u, Y64, YtY)
        if cg_steps is None:
            out[u] = np.linalg.solve(A, b)
        else:
            out[u] = cg_steps_f64(A, b, X[u], cg_steps)
    return out


def fit_f64(user_items: sp.csr_matrix, factors: int, regularization: float, alpha: float, iterations: int,
            random_state, cg_steps=3):
    Cui = sp.csr_matrix(user_items).astype(np.float64)
    if alpha != 1.0:
        Cui = sp.csr_matrix(Cui * alpha)
    Ciu = sp.csr_matrix(Cui.T)
    rs = np.random.RandomState(random_state)
    X = (rs.rand(Cui.shape[0], factors).astype(np.float32) * 0.01).astype(np.float64)
    Y = (rs.rand(Cui.shape[1], factors).astype(np.float32) * 0.01).astype(np.float64)
    for _ in range(iterations):
        X = half_sweep_f64(Cui, X, Y, regularization, cg_steps)
        Y = half_sweep_f64(Ciu, Y, X, regularization, cg_steps)
    return X, Y


def ranking_metrics(recommended: np.ndarray, held_out: sp.csr_matrix, k: int = 20):
    """precision / recall / F1 / hit-rate @k averaged over users that have
    held-out items (metric definition of the reference's evaluation cell,
    notebooks/baseline_als.ipynb cell 66ca94ea: per-user
    hits/len(recs[:k]), hits/len(actual), harmonic mean; hit-rate = any hit)."""
    held_out = sp.csr_matrix(held_out)
    P = R = F = H = 0.0
    n = 0
    for u in range(held_out.shape[0]):
        actual = set(held_out.indices[held_out.indptr[u]:held_out.indptr[u + 1]].tolist())
        if not actual:
            continue
        recs = [int(i) for i in recommended[u][:k] if i >= 0]
        hits = len(actual.intersection(recs))
        p = hits / len(recs) if recs else 0.0
        r = hits / len(actual)
        P += p
        R += r
        F += (2 * p * r / (p + r)) if (p + r) > 0 else 0.0
        H += 1.0 if hits > 0 else 0.0
        n += 1
    n = max(n, 1)
    return {"precision": P / n, "recall": R / n, "f1": F / n, "hit_rate": H / n, "users": n}
