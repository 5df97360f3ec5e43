"""ctypes front end of ``libmirec_oracle.so`` plus the fit driver restated from
implicit 0.7.2's ``AlternatingLeastSquares.fit`` (cpu/als.py), as the reference
calls it at ``als/als.py:167-174`` and
``src/shrekommender_system/recommenders/als_recommender.py:128-136``.

Test infrastructure only (see ``oracle/__init__.py``).  PARITY UNPINNED.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Callable, Optional, Tuple

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmirec_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


_LIB_V4_PATH = os.path.join(_HERE, "libmirec_oracle_v4.so")


def build(force: bool = False) -> str:
    """Compile the C restatement next to its source (gcc, a few seconds): the x86-64-v3 build every check uses
    and the x86-64-v4 build that is only ever timed."""
    src = os.path.join(_HERE, "mirec_oracle.c")
    for path in (_LIB_PATH, _LIB_V4_PATH):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(path)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def use_widest_build() -> str:
    """For timing the oracle as a CPU baseline: switch this process to the AVX-512 build when the host has
    AVX-512 and that build exists (results then differ from the x86-64-v3 build in the last bits, so no parity
    check may follow in the same process).  Returns the -march level in use."""
    global _lib
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if "avx512f" in flags and "avx512vl" in flags and "avx512bw" in flags and "avx512dq" in flags \
            and os.path.exists(_LIB_V4_PATH):
        _lib = ctypes.CDLL(_LIB_V4_PATH)
        _lib.oracle_als_loss.restype = ctypes.c_double
        _lib.oracle_num_threads.restype = ctypes.c_int
        return "x86-64-v4"
    _load()
    return "x86-64-v3"


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_als_loss.restype = ctypes.c_double
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _p(a: Optional[np.ndarray], typ):
    if a is None:
        return ctypes.cast(None, typ)
    return a.ctypes.data_as(typ)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _csr_parts(m: sp.csr_matrix):
    m = sp.csr_matrix(m)
    if not m.has_sorted_indices:
        m = m.sorted_indices()
    return (
        np.ascontiguousarray(m.indptr, dtype=np.int64),
        np.ascontiguousarray(m.indices, dtype=np.int32),
        np.ascontiguousarray(m.data, dtype=np.float32),
    )


def num_threads() -> int:
    return int(_load().oracle_num_threads())


def set_num_threads(n: int) -> None:
    _load().oracle_set_num_threads(ctypes.c_int(int(n)))


def gram(Y: np.ndarray, reg: float) -> np.ndarray:
    Y = _f32(Y)
    n, f = Y.shape
    out = np.empty((f, f), dtype=np.float32)
    _load().oracle_gram(_p(Y, _f32p), ctypes.c_int64(n), ctypes.c_int(f), ctypes.c_float(reg), _p(out, _f32p))
    return out


def als_half_sweep(C: sp.csr_matrix, X: np.ndarray, Y: np.ndarray, reg: float, cg_steps: int = 3,
                   rows: Optional[Tuple[int, int]] = None, YtY: Optional[np.ndarray] = None) -> None:
    """Update X in place from Y (implicit's `_least_squares_cg`)."""
    assert X.dtype == np.float32 and X.flags.c_contiguous
    Y = _f32(Y)
    indptr, indices, data = _csr_parts(C)
    f = X.shape[1]
    if YtY is None:
        YtY = gram(Y, reg)
    r0, r1 = rows if rows is not None else (0, X.shape[0])
    _load().oracle_als_half_sweep(
        _p(indptr, _i64p), _p(indices, _i32p), _p(data, _f32p), ctypes.c_int64(r0), ctypes.c_int64(r1),
        _p(X, _f32p), _p(Y, _f32p), _p(_f32(YtY), _f32p), ctypes.c_int(f), ctypes.c_int(cg_steps))


def als_half_sweep_exact(C: sp.csr_matrix, X: np.ndarray, Y: np.ndarray, reg: float) -> None:
    """Update X in place by the exact normal-equation solve (`use_cg=False`)."""
    assert X.dtype == np.float32 and X.flags.c_contiguous
    Y = _f32(Y)
    indptr, indices, data = _csr_parts(C)
    f = X.shape[1]
    YtY = gram(Y, reg)
    _load().oracle_als_half_sweep_exact(
        _p(indptr, _i64p), _p(indices, _i32p), _p(data, _f32p), ctypes.c_int64(0), ctypes.c_int64(X.shape[0]),
        _p(X, _f32p), _p(Y, _f32p), _p(YtY, _f32p), ctypes.c_int(f))


def als_loss(C: sp.csr_matrix, X: np.ndarray, Y: np.ndarray, reg: float) -> float:
    indptr, indices, data = _csr_parts(C)
    X, Y = _f32(X), _f32(Y)
    return float(_load().oracle_als_loss(
        _p(indptr, _i64p), _p(indices, _i32p), _p(data, _f32p), ctypes.c_int64(X.shape[0]),
        ctypes.c_int64(Y.shape[0]), _p(X, _f32p), _p(Y, _f32p), ctypes.c_int(X.shape[1]), ctypes.c_float(reg)))


def init_factors(n_users: int, n_items: int, factors: int, random_state: Optional[int]):
    """implicit's initialisation: one RandomState stream, users first."""
    rs = np.random.RandomState(random_state)
    X = rs.rand(n_users, factors).astype(np.float32) * 0.01
    Y = rs.rand(n_items, factors).astype(np.float32) * 0.01
    return np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(Y, dtype=np.float32)


def als_fit(user_items: sp.csr_matrix, factors: int = 100, regularization: float = 0.01, alpha: float = 1.0,
            iterations: int = 15, random_state: Optional[int] = None, cg_steps: int = 3, use_cg: bool = True,
            callback: Optional[Callable[[int, np.ndarray, np.ndarray], None]] = None):
    """Restatement of ``AlternatingLeastSquares.fit``: cast to fp32, scale by
    alpha when it is not 1, transpose once, draw X then Y from one
    ``RandomState(seed)`` stream, then ``iterations`` x {user sweep, item sweep}.
    Returns (user_factors, item_factors)."""
    Cui = sp.csr_matrix(user_items).astype(np.float32)
    if alpha != 1.0:
        Cui = Cui * np.float32(alpha)
        Cui = sp.csr_matrix(Cui).astype(np.float32)
    Ciu = sp.csr_matrix(Cui.T)
    users, items = Cui.shape
    X, Y = init_factors(users, items, factors, random_state)
    for it in range(iterations):
        if use_cg:
            als_half_sweep(Cui, X, Y, regularization, cg_steps)
            als_half_sweep(Ciu, Y, X, regularization, cg_steps)
        else:
            als_half_sweep_exact(Cui, X, Y, regularization)
            als_half_sweep_exact(Ciu, Y, X, regularization)
        if callback is not None:
            callback(it, X, Y)
    return X, Y


def scores(Q: np.ndarray, Y: np.ndarray) -> np.ndarray:
    Q, Y = _f32(Q), _f32(Y)
    out = np.empty((Q.shape[0], Y.shape[0]), dtype=np.float32)
    _load().oracle_scores(_p(Q, _f32p), ctypes.c_int64(Q.shape[0]), _p(Y, _f32p), ctypes.c_int64(Y.shape[0]),
                          ctypes.c_int(Q.shape[1]), _p(out, _f32p))
    return out


def topk(Q: np.ndarray, Y: np.ndarray, k: int, liked: Optional[sp.csr_matrix] = None,
         y_norm: Optional[np.ndarray] = None, q_norm: Optional[np.ndarray] = None):
    """Top-k items per query row; `liked` (one CSR row per query) is filtered."""
    Q, Y = _f32(Q), _f32(Y)
    nq, f = Q.shape
    ni = Y.shape[0]
    ids = np.empty((nq, k), dtype=np.int32)
    sc = np.empty((nq, k), dtype=np.float32)
    if liked is not None:
        indptr, indices, _ = _csr_parts(liked)
        assert liked.shape[0] == nq
    else:
        indptr = indices = None
    yn = None if y_norm is None else _f32(y_norm)
    qn = None if q_norm is None else _f32(q_norm)
    _load().oracle_topk(_p(Q, _f32p), ctypes.c_int64(nq), _p(Y, _f32p), ctypes.c_int64(ni), ctypes.c_int(f),
                        _p(indptr, _i64p), _p(indices, _i32p), ctypes.c_int(k), _p(yn, _f32p), _p(qn, _f32p),
                        _p(ids, _i32p), _p(sc, _f32p))
    return ids, sc


def lightfm_repr(features: sp.csr_matrix, embeddings: np.ndarray, biases: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """[rows, nc+1] representations; feature rows are summed in stored order."""
    m = sp.csr_matrix(features)
    indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(m.indices, dtype=np.int32)
    w = np.ascontiguousarray(m.data, dtype=np.float32)
    E, b = _f32(embeddings), _f32(biases)
    nc = E.shape[1]
    out = np.empty((m.shape[0], nc + 1), dtype=np.float32)
    _load().oracle_lightfm_repr(_p(indptr, _i64p), _p(indices, _i32p), _p(w, _f32p), ctypes.c_int64(m.shape[0]),
                                _p(E, _f32p), _p(b, _f32p), ctypes.c_float(scale), ctypes.c_int(nc), _p(out, _f32p))
    return out


def lightfm_predict(urepr: np.ndarray, irepr: np.ndarray, user_ids, item_ids) -> np.ndarray:
    urepr, irepr = _f32(urepr), _f32(irepr)
    u = np.ascontiguousarray(user_ids, dtype=np.int32)
    i = np.ascontiguousarray(item_ids, dtype=np.int32)
    nc = urepr.shape[1] - 1
    out = np.empty(len(u), dtype=np.float32)
    _load().oracle_lightfm_predict(_p(urepr, _f32p), _p(irepr, _f32p), _p(u, _i32p), _p(i, _i32p),
                                   ctypes.c_int64(len(u)), ctypes.c_int(nc), _p(out, _f32p))
    return out


def lightfm_topk(urepr: np.ndarray, irepr: np.ndarray, k: int):
    urepr, irepr = _f32(urepr), _f32(irepr)
    nc = urepr.shape[1] - 1
    ids = np.empty((urepr.shape[0], k), dtype=np.int32)
    sc = np.empty((urepr.shape[0], k), dtype=np.float32)
    _load().oracle_lightfm_topk(_p(urepr, _f32p), ctypes.c_int64(urepr.shape[0]), _p(irepr, _f32p),
                                ctypes.c_int64(irepr.shape[0]), ctypes.c_int(nc), ctypes.c_int(k),
                                _p(ids, _i32p), _p(sc, _f32p))
    return ids, sc


def lightfm_ranks(urepr: np.ndarray, irepr: np.ndarray, test: sp.csr_matrix, train: Optional[sp.csr_matrix] = None) -> np.ndarray:
    """lightfm `predict_rank`: one float per stored test entry (CSR order)."""
    urepr, irepr = _f32(urepr), _f32(irepr)
    nc = urepr.shape[1] - 1
    t = sp.csr_matrix(test)
    t.sort_indices()
    tp = np.ascontiguousarray(t.indptr, dtype=np.int64)
    ti = np.ascontiguousarray(t.indices, dtype=np.int32)
    rp = ri = None
    if train is not None:
        r = sp.csr_matrix(train)
        r.sort_indices()
        rp = np.ascontiguousarray(r.indptr, dtype=np.int64)
        ri = np.ascontiguousarray(r.indices, dtype=np.int32)
    ranks = np.zeros(t.nnz, dtype=np.float32)
    _load().oracle_lightfm_ranks(_p(urepr, _f32p), _p(irepr, _f32p), ctypes.c_int64(t.shape[0]), ctypes.c_int64(t.shape[1]),
                                 ctypes.c_int(nc), _p(tp, _i64p), _p(ti, _i32p), _p(rp, _i64p), _p(ri, _i32p),
                                 _p(ranks, _f32p))
    return ranks


def lightfm_auc(ranks: np.ndarray, test: sp.csr_matrix, train: Optional[sp.csr_matrix] = None,
                preserve_rows: bool = False) -> np.ndarray:
    """lightfm.evaluation.auc_score from the ranks of `lightfm_ranks`."""
    t = sp.csr_matrix(test)
    tp = np.ascontiguousarray(t.indptr, dtype=np.int64)
    ntp = (np.zeros(t.shape[0], dtype=np.int32) if train is None
           else np.ascontiguousarray(np.diff(sp.csr_matrix(train).indptr), dtype=np.int32))
    data = np.array(ranks, dtype=np.float32, copy=True)
    auc = np.zeros(t.shape[0], dtype=np.float32)
    _load().oracle_lightfm_auc_from_ranks(_p(tp, _i64p), ctypes.c_int64(t.shape[0]), ctypes.c_int64(t.shape[1]),
                                          _p(ntp, _i32p), _p(data, _f32p), _p(auc, _f32p))
    return auc if preserve_rows else auc[np.diff(t.indptr) > 0]


def lightfm_init(n_user_features: int, n_item_features: int, no_components: int, random_state: int):
    """LightFM._initialize (lightfm/lightfm.py) for the adagrad schedule: embeddings
    (rand - 0.5) / no_components in float32, items first; biases zero; squared-gradient
    accumulators one."""
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    nc = no_components
    st = {
        "item_embeddings": ((rs.rand(n_item_features, nc) - 0.5) / nc).astype(np.float32),
        "item_embedding_gradients": np.ones((n_item_features, nc), dtype=np.float32),
        "item_biases": np.zeros(n_item_features, dtype=np.float32),
        "item_bias_gradients": np.ones(n_item_features, dtype=np.float32),
    }
    st.update({
        "user_embeddings": ((rs.rand(n_user_features, nc) - 0.5) / nc).astype(np.float32),
        "user_embedding_gradients": np.ones((n_user_features, nc), dtype=np.float32),
        "user_biases": np.zeros(n_user_features, dtype=np.float32),
        "user_bias_gradients": np.ones(n_user_features, dtype=np.float32),
    })
    return st


def lightfm_warp_epoch(state: dict, interactions: sp.coo_matrix, user_features: sp.csr_matrix,
                       item_features: sp.csr_matrix, order: np.ndarray, seed: int, learning_rate: float = 0.05,
                       item_alpha: float = 0.0, user_alpha: float = 0.0, max_sampled: int = 10,
                       sample_weight: Optional[np.ndarray] = None) -> None:
    """One sequential WARP epoch over the examples of `interactions` (COO) in `order`; `state` is
    updated in place."""
    coo = sp.coo_matrix(interactions)
    uf, itf = sp.csr_matrix(user_features), sp.csr_matrix(item_features)
    pos = sp.csr_matrix(interactions)
    pos.sort_indices()
    nc = state["item_embeddings"].shape[1]
    a = lambda x, t: np.ascontiguousarray(x, dtype=t)
    ufp, ufi, ufd = a(uf.indptr, np.int64), a(uf.indices, np.int32), a(uf.data, np.float32)
    ifp, ifi, ifd = a(itf.indptr, np.int64), a(itf.indices, np.int32), a(itf.data, np.float32)
    pp, pi = a(pos.indptr, np.int64), a(pos.indices, np.int32)
    uid, iid, y = a(coo.row, np.int32), a(coo.col, np.int32), a(coo.data, np.float32)
    sw = None if sample_weight is None else a(sample_weight, np.float32)
    od = a(order, np.int32)
    scales = np.array([1.0, 1.0], dtype=np.float64)
    for k in state:
        assert state[k].dtype == np.float32 and state[k].flags.c_contiguous
    _load().oracle_lightfm_warp_epoch(
        _p(ifp, _i64p), _p(ifi, _i32p), _p(ifd, _f32p), ctypes.c_int64(itf.shape[0]),
        _p(ufp, _i64p), _p(ufi, _i32p), _p(ufd, _f32p), _p(pp, _i64p), _p(pi, _i32p),
        _p(uid, _i32p), _p(iid, _i32p), _p(y, _f32p), _p(sw, _f32p), _p(od, _i32p), ctypes.c_int64(len(od)),
        _p(state["item_embeddings"], _f32p), _p(state["item_embedding_gradients"], _f32p),
        _p(state["item_biases"], _f32p), _p(state["item_bias_gradients"], _f32p), ctypes.c_int64(itf.shape[1]),
        _p(state["user_embeddings"], _f32p), _p(state["user_embedding_gradients"], _f32p),
        _p(state["user_biases"], _f32p), _p(state["user_bias_gradients"], _f32p), ctypes.c_int64(uf.shape[1]),
        ctypes.c_int(nc), ctypes.c_double(learning_rate), ctypes.c_double(item_alpha), ctypes.c_double(user_alpha),
        ctypes.c_int(max_sampled), ctypes.c_uint32(seed & 0xffffffff), scales.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
