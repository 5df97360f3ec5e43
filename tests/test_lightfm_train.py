"""LightFM WARP training (`fit_partial` as called at hybrid_docker/utils.py:100-107).

The library's training is Hogwild (unsynchronised OpenMP threads, rand_r
negatives): no two runs of it agree bit for bit, and lightfm 1.17 is not
installed here, so the sequential CPU statement in oracle/ is itself unpinned.
What is checked: the statement learns a planted structure; the device kernel in
serial mode reproduces the statement exactly (same order, same generator); the
concurrent device epochs reach the statement's held-out AUC within 0.02."""

import numpy as np
import pytest
import scipy.sparse as sp

import oracle


def planted(seed=0, nu=1500, ni=500, clusters=8, per_user=25):
    from movie_recommender_system_amd import synth
    p = synth.lightfm_problem(9, nu, ni)
    rng = np.random.default_rng(seed)
    cu, ci = rng.integers(0, clusters, nu), rng.integers(0, clusters, ni)
    rows, cols = [], []
    for u in range(nu):
        cand = np.where(ci == cu[u])[0]
        pick = rng.choice(cand, size=min(len(cand), per_user), replace=False)
        for i in np.unique(np.r_[pick, rng.integers(0, ni, 3)]):
            rows.append(u)
            cols.append(i)
    inter = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(nu, ni))
    train, test = synth.split_holdout(inter, 3, 0.2)
    return p, sp.coo_matrix(train), sp.csr_matrix(test)


def oracle_auc(st, p, train, test):
    ur = oracle.lightfm_repr(p["user_features"], st["user_embeddings"], st["user_biases"])
    ir = oracle.lightfm_repr(p["item_features"], st["item_embeddings"], st["item_biases"])
    tr = sp.csr_matrix(train)
    return float(oracle.lightfm_auc(oracle.lightfm_ranks(ur, ir, test, tr), test, tr).mean())


def oracle_fit(p, train, epochs, item_alpha, seed=42):
    rs = np.random.RandomState(seed)
    st = oracle.lightfm_init(p["user_features"].shape[1], p["item_features"].shape[1], 15, rs)
    for _ in range(epochs):
        order = np.arange(train.nnz, dtype=np.int32)
        rs.shuffle(order)
        s = int(rs.randint(0, np.iinfo(np.int32).max))
        oracle.lightfm_warp_epoch(st, train, p["user_features"], p["item_features"], order, s, item_alpha=item_alpha)
    return st


def test_sequential_statement_learns_planted_structure():
    p, train, test = planted()
    aucs = []
    rs = np.random.RandomState(42)
    st = oracle.lightfm_init(p["user_features"].shape[1], p["item_features"].shape[1], 15, rs)
    aucs.append(oracle_auc(st, p, train, test))
    for _ in range(4):
        order = np.arange(train.nnz, dtype=np.int32)
        rs.shuffle(order)
        oracle.lightfm_warp_epoch(st, train, p["user_features"], p["item_features"], order,
                                  int(rs.randint(0, 2**31 - 1)), item_alpha=1e-6)
        aucs.append(oracle_auc(st, p, train, test))
    assert abs(aucs[0] - 0.5) < 0.05 and aucs[-1] > 0.85 and aucs[2] > aucs[1] > aucs[0]
    assert all(np.isfinite(v).all() for v in st.values())
    assert (st["item_embedding_gradients"] >= 1).all() and st["user_bias_gradients"].max() == 1.0  # WARP leaves user biases alone


def test_model_rejects_what_is_not_built():
    from movie_recommender_system_amd import LightFM
    with pytest.raises(NotImplementedError):
        LightFM(loss="bpr")
    with pytest.raises(NotImplementedError):
        LightFM(loss="warp", learning_schedule="adadelta")
    m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=1e-6)
    assert m.no_components == 15 and m.max_sampled == 10 and m.learning_rate == 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("item_alpha", [0.0, 1e-3])
def test_device_serial_epoch_equals_the_sequential_statement(item_alpha):
    import torch
    assert torch.cuda.is_available()
    from movie_recommender_system_amd import LightFM
    p, train, test = planted(nu=300, ni=200, per_user=12)
    st = oracle_fit(p, train, 2, item_alpha)
    m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=item_alpha,
                serial=True)
    for _ in range(2):  # exactly the reference's loop: one epoch per call (hybrid_docker/utils.py:97-107)
        m.fit_partial(train, user_features=p["user_features"], item_features=p["item_features"], epochs=1, num_threads=4)
    for name in ("item_embeddings", "item_biases", "user_embeddings", "user_biases", "item_embedding_gradients",
                 "item_bias_gradients", "user_embedding_gradients", "user_bias_gradients"):
        assert np.array_equal(getattr(m, name), st[name]), name


@pytest.mark.gpu
def test_device_concurrent_epochs_reach_the_statements_auc():
    import torch
    assert torch.cuda.is_available()
    from movie_recommender_system_amd import LightFM
    from movie_recommender_system_amd.lightfm import auc_score
    p, train, test = planted()
    st = oracle_fit(p, train, 5, 1e-6)
    want = oracle_auc(st, p, train, test)
    m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=1e-6)
    m.fit_partial(train, user_features=p["user_features"], item_features=p["item_features"], epochs=5, num_threads=4)
    got = float(auc_score(m, test, sp.csr_matrix(train), user_features=p["user_features"],
                          item_features=p["item_features"], num_threads=4).mean())
    assert want > 0.85 and abs(got - want) < 0.02, (got, want)
    # the fitted model serves like a loaded one (hybrid_docker/inference.py:159-165)
    s = m.predict(3, np.arange(10, dtype=np.int32), user_features=p["user_features"], item_features=p["item_features"])
    assert s.shape == (10,) and np.isfinite(s).all()


@pytest.mark.gpu
def test_randomised_lightfm_parity_fixed_seed_and_empty_epoch():
    """Fixed-seed slice of the randomised differential tool: 150 LightFM scoring / ranking cases and 100
    serial WARP fits over random sizes, component counts, feature matrices, weights and penalties, all bit
    for bit; an epoch over no interactions leaves the model as initialised."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for i in range(150):
        ok, desc = fz.case_lightfm_eval(np.random.default_rng([21, i]))
        assert ok, f"case {i}: {desc}"
    for i in range(100):
        ok, desc = fz.case_warp(np.random.default_rng([22, i]))
        assert ok, f"case {i}: {desc}"
    from movie_recommender_system_amd import LightFM
    m = LightFM(no_components=4, loss="warp", random_state=1, learning_schedule="adagrad", serial=True)
    m.fit_partial(sp.coo_matrix((5, 7), dtype=np.float32), epochs=1)
    st = oracle.lightfm_init(5, 7, 4, 1)
    assert all(np.array_equal(getattr(m, k), v) for k, v in st.items())


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_concurrent_epochs_keep_learning_at_scale():
    """10 k users x 1.5 k items, 0.2 M examples: thousands of examples in flight from the first one on learn
    next to nothing (every adagrad accumulator starts at one and the feature rows all users share take thousands
    of full-size steps at once); the phased schedule of the device epoch must stay within 0.02 of the serial
    mode's held-out AUC after three epochs."""
    import torch
    assert torch.cuda.is_available()
    from movie_recommender_system_amd import LightFM
    from movie_recommender_system_amd.lightfm import auc_score
    p, train, test = planted(seed=1, nu=10000, ni=1500, clusters=20, per_user=25)
    tr = sp.csr_matrix(train)
    aucs = {}
    for serial in (False, True):
        m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=1e-6,
                    serial=serial)
        m.fit_partial(train, user_features=p["user_features"], item_features=p["item_features"], epochs=3)
        aucs[serial] = float(auc_score(m, test, tr, user_features=p["user_features"],
                                       item_features=p["item_features"]).mean())
    assert aucs[True] > 0.85 and abs(aucs[False] - aucs[True]) < 0.02, aucs


def test_model_object_round_trips_through_joblib(tmp_path):
    """hybrid_docker/train.py:182 dumps the model object with joblib and hybrid_docker/inference.py:102 loads it
    again: parameters, accumulators, hyper-parameters and the random state survive; the lock and the device
    tables are rebuilt."""
    import joblib
    from movie_recommender_system_amd import LightFM, LightFMScorer
    m = LightFM(no_components=6, loss="warp", random_state=7, learning_schedule="adagrad", item_alpha=1e-6)
    m._initialize(9, 11)
    m._examples_seen = 1234
    path = tmp_path / "lightfm_model.joblib"
    joblib.dump(m, path)
    r = joblib.load(path)
    for name in ("item_embeddings", "item_biases", "user_embeddings", "user_biases", "item_embedding_gradients",
                 "item_bias_gradients", "user_embedding_gradients", "user_bias_gradients"):
        assert np.array_equal(getattr(r, name), getattr(m, name)), name
    assert (r.no_components, r.item_alpha, r.max_sampled, r.learning_rate, r._examples_seen) == (6, 1e-6, 10, 0.05, 1234)
    assert r.random_state.randint(1 << 30) == m.random_state.randint(1 << 30)
    assert r._lock is not m._lock and r._dev == {}
    sc = LightFMScorer(m.user_embeddings, m.user_biases, m.item_embeddings, m.item_biases)
    joblib.dump(sc, path)
    assert np.array_equal(joblib.load(path).item_embeddings, m.item_embeddings)
