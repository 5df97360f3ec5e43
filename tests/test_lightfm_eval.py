"""LightFM evaluation (`predict_rank`, `auc_score`, `precision_at_k`, `recall_at_k`
as called at hybrid_docker/utils.py:111-155): the oracle against hand-computed
cases on the CPU, the device kernel against the oracle on the GPU (ranks are
integer counts: equal exactly; AUC: float32 terms, 1e-6).  lightfm 1.17 is not
installed here, so the statement followed (`predict_ranks`,
`calculate_auc_from_rank` of its `_lightfm_fast`) is itself unpinned."""

import numpy as np
import pytest
import scipy.sparse as sp

import oracle


def _tiny():
    # one component, no biases: scores = u * i
    urepr = np.array([[1.0, 0.0], [-1.0, 0.0], [1.0, 0.0]], dtype=np.float32)
    irepr = np.array([[5.0, 0.0], [4.0, 0.0], [3.0, 0.0], [3.0, 0.0], [1.0, 0.0]], dtype=np.float32)
    test = sp.csr_matrix(np.array([[0, 1, 0, 1, 0], [1, 0, 0, 0, 1], [0, 0, 0, 0, 0]], dtype=np.float32))
    train = sp.csr_matrix(np.array([[1, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 1, 0, 0, 0]], dtype=np.float32))
    return urepr, irepr, test, train


def test_oracle_ranks_and_auc_hand_computed():
    urepr, irepr, test, train = _tiny()
    r = oracle.lightfm_ranks(urepr, irepr, test, train)
    # user 0 (scores 5 4 3 3 1, item 0 excluded): item 1 -> nobody above: 0; item 3 -> items 1 and 2 (tie counts): 2
    # user 1 (scores -5 -4 -3 -3 -1): item 0 -> all four others: 4; item 4 -> 0
    assert r.tolist() == [0.0, 2.0, 4.0, 0.0]
    r_all = oracle.lightfm_ranks(urepr, irepr, test, None)
    assert r_all.tolist() == [1.0, 3.0, 4.0, 0.0]
    auc = oracle.lightfm_auc(r, test, train, preserve_rows=True)
    # user 0: negatives = 5 - (2 + 1) = 2; sorted ranks 0, 2 -> adjusted 0, 1 -> (1 + 0.5) / 2
    # user 1: negatives = 3; sorted ranks 0, 4 -> adjusted 0, 3 -> (1 + 0) / 2;  user 2: no positives -> 0.5
    assert np.allclose(auc, [0.75, 0.5, 0.5], atol=1e-7)
    assert np.allclose(oracle.lightfm_auc(r, test, train), [0.75, 0.5], atol=1e-7)


def test_host_auc_from_ranks_matches_oracle():
    from movie_recommender_system_amd.lightfm import auc_from_ranks
    rng = np.random.default_rng(3)
    nu, ni = 300, 90
    test = sp.random(nu, ni, density=0.08, random_state=np.random.RandomState(1), format="csr", dtype=np.float32)
    test.sort_indices()
    ntp = rng.integers(0, 20, nu).astype(np.int32)
    ranks = test.copy()
    ranks.data = rng.integers(0, ni - 25, test.nnz).astype(np.float32)
    got = auc_from_ranks(ranks, ntp)
    data = ranks.data.copy()
    # the oracle takes the train matrix only for its row counts
    train = sp.csr_matrix((np.ones(int(ntp.sum()), dtype=np.float32),
                           np.concatenate([np.arange(n) for n in ntp]).astype(np.int32),
                           np.r_[0, np.cumsum(ntp)]), shape=(nu, ni))
    want = oracle.lightfm_auc(data, test, train, preserve_rows=True)
    assert np.allclose(got, want, atol=1e-6)


@pytest.mark.gpu
def test_device_ranks_auc_precision_recall_match_oracle():
    import torch
    assert torch.cuda.is_available()
    from movie_recommender_system_amd import LightFMScorer, synth
    from movie_recommender_system_amd.lightfm import auc_score, precision_at_k, recall_at_k
    nu, ni = 900, 777
    p = synth.lightfm_problem(9, nu, ni)
    inter = synth.interactions(5, nu, ni, 30.0)
    train, test = synth.split_holdout(inter, 7, 0.25)
    test = sp.csr_matrix(test)
    test[5] = 0   # a user without test interactions
    test.eliminate_zeros()
    # a user with more than 64 test interactions (two passes of the kernel) and exact score ties (duplicated items)
    dense = test.toarray()
    dense[11, :200] = 1
    test = sp.csr_matrix(dense.astype(np.float32))
    train = sp.csr_matrix(train.multiply(test == 0))
    train.eliminate_zeros()
    p["item_embeddings"][ni + 3] = p["item_embeddings"][ni + 4]
    sc = LightFMScorer(p["user_embeddings"], p["user_biases"], p["item_embeddings"], p["item_biases"])
    ur = oracle.lightfm_repr(p["user_features"], p["user_embeddings"], p["user_biases"])
    ir = oracle.lightfm_repr(p["item_features"], p["item_embeddings"], p["item_biases"])
    for tr in (train, None):
        got = sc.predict_rank(test, tr, user_features=p["user_features"], item_features=p["item_features"])
        want = oracle.lightfm_ranks(ur, ir, test, tr)
        assert got.shape == test.shape and np.array_equal(got.indptr, test.indptr)
        assert np.array_equal(got.data, want)
        auc = auc_score(sc, test, tr, user_features=p["user_features"], item_features=p["item_features"])
        assert auc.shape == (int((np.diff(test.indptr) > 0).sum()),) and auc.dtype == np.float32
        assert auc.shape[0] < nu
        assert np.allclose(auc, oracle.lightfm_auc(want, test, tr), atol=1e-6)
    # precision / recall at 20 exactly as hybrid_docker/utils.py:119-131 calls them
    prec = precision_at_k(sc, test, train, k=20, user_features=p["user_features"], item_features=p["item_features"])
    rec = recall_at_k(sc, test, train, k=20, user_features=p["user_features"], item_features=p["item_features"])
    want = oracle.lightfm_ranks(ur, ir, test, train)
    hit = np.add.reduceat(np.r_[(want < 20).astype(np.float64), 0.0], test.indptr[:-1])[np.diff(test.indptr) > 0]
    assert np.allclose(prec, hit / 20) and np.allclose(rec, hit / np.diff(test.indptr)[np.diff(test.indptr) > 0])
    with pytest.raises(ValueError):
        sc.predict_rank(test, test, user_features=p["user_features"], item_features=p["item_features"])
