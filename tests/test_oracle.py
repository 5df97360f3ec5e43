"""CPU checks of the oracle itself (no GPU): golden vectors, an independent
float64 statement of the same algorithm, and the invariants of the method.

The reference pins no numeric result for this path (its tests mock implicit /
lightfm), so the fixtures in tests/golden were produced by this oracle
(tests/golden/make_golden.py): they pin the build against regressions, not
against the third-party libraries -- "parity unpinned".
"""

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from oracle import als_numpy
from movie_recommender_system_amd import synth


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def small_matrix(seed=1, rows=200, cols=80, density=0.05):
    M = sp.random(rows, cols, density=density, random_state=np.random.RandomState(seed), format="csr",
                  dtype=np.float32)
    M.data[:] = 1.0
    return M


def test_golden_reference_fixture(golden_dir):
    """The reference's 2x3 test matrix (tests/test_als_recommender.py:13), f=2, 1 iteration, seed 42."""
    M = sp.csr_matrix(np.array([[1, 0, 1], [0, 1, 0]], dtype=np.float32))
    X, Y = oracle.als_fit(M, factors=2, regularization=0.01, alpha=40, iterations=1, random_state=42)
    g = np.load(f"{golden_dir}/kat1.npz")
    np.testing.assert_allclose(X, g["X"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(Y, g["Y"], rtol=1e-5, atol=1e-7)
    X64, Y64 = als_numpy.fit_f64(M, 2, 0.01, 40, 1, 42)
    assert rel(X, X64) < 1e-5 and rel(Y, Y64) < 1e-5


def test_golden_random_case_and_topk(golden_dir):
    M = small_matrix()
    X, Y = oracle.als_fit(M, factors=16, regularization=0.01, alpha=40, iterations=5, random_state=42)
    g = np.load(f"{golden_dir}/kat2.npz")
    assert rel(X, g["X"]) < 1e-5 and rel(Y, g["Y"]) < 1e-5
    ids, sc = oracle.topk(g["X"], g["Y"], 20, liked=M)
    assert np.array_equal(ids, g["top_ids"].astype(np.int32))
    assert np.array_equal(sc, g["top_scores"])
    X64, Y64 = als_numpy.fit_f64(M, 16, 0.01, 40, 5, 42)
    assert rel(X, X64) < 5e-5 and rel(Y, Y64) < 5e-5


def test_golden_zipf_case(golden_dir):
    full = synth.interactions(45, 2000, 500, 30.0)
    train, held = synth.split_holdout(full, 46, 0.2)
    g = np.load(f"{golden_dir}/kat3.npz")
    assert train.nnz == int(g["train_nnz"]) and held.nnz == int(g["held_nnz"])
    X, Y = oracle.als_fit(train, factors=64, regularization=0.01, alpha=40, iterations=15, random_state=42)
    assert rel(Y, g["Y"]) < 1e-4 and rel(X[::8], g["X_every8"]) < 1e-4
    ids, _ = oracle.topk(X, Y, 20, liked=train)
    assert (ids == g["top_ids"].astype(np.int32)).mean() > 0.999
    m = als_numpy.ranking_metrics(ids, held, 20)
    assert 0.0 < m["precision"] < 1.0 and m["hit_rate"] > 0.1


def test_golden_lightfm(golden_dir):
    p = synth.lightfm_problem(5, 200, 90)
    ur = oracle.lightfm_repr(p["user_features"], p["user_embeddings"], p["user_biases"])
    ir = oracle.lightfm_repr(p["item_features"], p["item_embeddings"], p["item_biases"])
    g = np.load(f"{golden_dir}/lightfm_kat.npz")
    assert np.array_equal(ur, g["user_repr"]) and np.array_equal(ir, g["item_repr"])
    ids, sc = oracle.lightfm_topk(ur, ir, 20)
    assert np.array_equal(ids, g["top_ids"].astype(np.int32)) and np.array_equal(sc, g["top_scores"])
    # against plain numpy: repr = features @ [E | b], score = bias sum + dot, argsort()[::-1]
    ur_np = p["user_features"] @ np.c_[p["user_embeddings"], p["user_biases"]]
    ir_np = p["item_features"] @ np.c_[p["item_embeddings"], p["item_biases"]]
    np.testing.assert_allclose(ur, ur_np, rtol=1e-5, atol=1e-6)
    scores = ur_np[:, :-1] @ ir_np[:, :-1].T + ur_np[:, -1:] + ir_np[:, -1]
    got = oracle.lightfm_predict(ur, ir, np.full(90, 7, dtype=np.int32), np.arange(90, dtype=np.int32))
    np.testing.assert_allclose(got, scores[7], rtol=1e-4, atol=1e-5)
    top = np.argsort(scores[7])[::-1][:5]
    assert set(top.tolist()) == set(ids[7][:5].tolist())


def test_many_cg_steps_equal_exact_solve():
    """With cg_steps >= f the CG iterates reach the normal-equation solution."""
    M = small_matrix(3, 120, 60, 0.1)
    Xa, Ya = oracle.als_fit(M, factors=8, regularization=0.01, alpha=40, iterations=2, random_state=1, cg_steps=24)
    Xb, Yb = oracle.als_fit(M, factors=8, regularization=0.01, alpha=40, iterations=2, random_state=1, use_cg=False)
    assert rel(Xa, Xb) < 1e-5 and rel(Ya, Yb) < 1e-5
    Xc, Yc = als_numpy.fit_f64(M, 8, 0.01, 40, 2, 1, cg_steps=None)
    assert rel(Xb, Xc) < 1e-5 and rel(Yb, Yc) < 1e-5


def test_loss_non_increasing_per_half_sweep_exact():
    M = small_matrix(5, 150, 70, 0.08)
    C = sp.csr_matrix(M * np.float32(40.0))
    Ct = sp.csr_matrix(C.T)
    X, Y = oracle.oracle.init_factors(150, 70, 12, 42)
    last = oracle.als_loss(C, X, Y, 0.01)
    for _ in range(4):
        oracle.als_half_sweep_exact(C, X, Y, 0.01)
        cur = oracle.als_loss(C, X, Y, 0.01)
        assert cur <= last * (1 + 1e-6)
        last = cur
        oracle.als_half_sweep_exact(Ct, Y, X, 0.01)
        cur = oracle.als_loss(C, X, Y, 0.01)
        assert cur <= last * (1 + 1e-6)
        last = cur


def test_empty_rows_are_zeroed_and_negative_confidence():
    rng = np.random.default_rng(0)
    M = sp.lil_matrix((6, 5), dtype=np.float32)
    M[0, 1] = 3.0
    M[0, 3] = -2.0     # "disliked": confidence 2, preference 0
    M[2, 0] = 1.0
    M[5, 4] = 7.0
    C = sp.csr_matrix(M)
    X = rng.random((6, 4)).astype(np.float32)
    Y = rng.standard_normal((5, 4)).astype(np.float32)
    Xo = X.copy()
    oracle.als_half_sweep(C, Xo, Y, 0.1, 3)
    assert np.all(Xo[[1, 3, 4]] == 0.0)
    X64 = als_numpy.half_sweep_f64(C, X, Y, 0.1, 3)
    assert rel(Xo, X64) < 1e-4
    # a row range leaves the other rows alone
    Xr = X.copy()
    oracle.als_half_sweep(C, Xr, Y, 0.1, 3, rows=(2, 4))
    assert np.array_equal(Xr[:2], X[:2]) and np.array_equal(Xr[4:], X[4:]) and np.array_equal(Xr[2:4], Xo[2:4])


def test_topk_order_and_filter():
    Q = np.array([[1.0, 0.0], [0.0, 0.0]], dtype=np.float32)
    Y = np.array([[3, 0], [5, 0], [5, 0], [1, 0], [4, 0]], dtype=np.float32)
    liked = sp.csr_matrix(np.array([[0, 1, 0, 0, 0], [0, 0, 0, 0, 0]], dtype=np.float32))
    ids, sc = oracle.topk(Q, Y, 3, liked=liked)
    assert ids[0].tolist() == [2, 4, 0] and sc[0].tolist() == [5.0, 4.0, 3.0]
    assert ids[1].tolist() == [0, 1, 2] and sc[1].tolist() == [0.0, 0.0, 0.0]      # ties: ascending index
    ids, sc = oracle.topk(Q, Y, 7, liked=liked)
    assert ids[0].tolist() == [2, 4, 0, 3, 1, -1, -1]
    assert sc[0][4] == -np.finfo(np.float32).max
    dense = oracle.scores(Q, Y)
    assert dense.shape == (2, 5) and dense[0].tolist() == [3.0, 5.0, 5.0, 1.0, 4.0]


def test_ranking_metrics_hand_case():
    held = sp.csr_matrix(np.array([[0, 1, 1, 0], [0, 0, 0, 0], [1, 0, 0, 0]], dtype=np.float32))
    recs = np.array([[1, 3], [0, 1], [2, 3]])
    m = als_numpy.ranking_metrics(recs, held, 2)
    # user 0: 1 hit of 2 recs, 2 actual -> p .5 r .5 f1 .5; user 1 skipped; user 2: 0 hits
    assert m["users"] == 2
    assert m["precision"] == pytest.approx(0.25) and m["recall"] == pytest.approx(0.25)
    assert m["f1"] == pytest.approx(0.25) and m["hit_rate"] == pytest.approx(0.5)
