"""``AlternatingLeastSquares``: the model object the reference builds with
``implicit.als.AlternatingLeastSquares(...)`` (``als/als.py:167-173``,
``recommenders/als_recommender.py:128-135``) and then drives through
``fit`` (``als/als.py:174``, ``als_recommender.py:136``), ``recommend``
(``als_recommender.py:181-186``), ``similar_items`` (``:237``) and
``__getstate__`` (``als/als.py:239``).

Same constructor arguments, attribute names and return conventions as
implicit 0.7.2's CPU model, so an instance can sit in ``als_model.pkl["model"]``
and be used by the serving code unchanged.  All arithmetic runs on the GPU
through libmirec; there is no CPU path.  State is held as plain numpy arrays so
that the pickle needs nothing but this package to load.
"""

from __future__ import annotations

import threading
from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from . import engine as E


def _check_csr(user_items) -> sp.csr_matrix:
    if not sp.isspmatrix_csr(user_items):
        user_items = sp.csr_matrix(user_items)
    if user_items.dtype != np.float32:
        user_items = user_items.astype(np.float32)
    return user_items


_FILTERED = float(np.finfo(np.float32).min)   # -FLT_MAX: the score of a liked / filtered item


def _hide_filtered(ids: torch.Tensor, scores: torch.Tensor, mask) -> torch.Tensor:
    """A user whose filter leaves fewer than N items gets the filtered ones back
    at the tail of the list with score -FLT_MAX: their ids are replaced by -1
    so that no caller recommends an item the user already has (the plugin drops
    negative ids, ``recommender.ALSRecommender._recommend_for_known_user``)."""
    if mask is None:
        return ids
    return torch.where(scores == _FILTERED, torch.full_like(ids, -1), ids)


class AlternatingLeastSquares:
    """Implicit-feedback ALS (Hu, Koren, Volinsky) with the conjugate-gradient
    solver of Takacs, Pilaszy and Tikk: ``cg_steps`` (3) warm-started steps per
    row per half-sweep.  ``use_cg=False`` selects implicit's exact solver
    instead (``_least_squares``: the normal equations of every row by
    Cholesky); it needs every stored value times ``alpha`` to be at least 1 in
    magnitude, which keeps the row systems positive definite."""

    SOLVE_EXACT = -1   # MIREC_SOLVE_EXACT of the C ABI

    def __init__(self, factors: int = 100, regularization: float = 0.01, alpha: float = 1.0, dtype=np.float32,
                 use_native: bool = True, use_cg: bool = True, use_gpu: bool = True, iterations: int = 15,
                 calculate_training_loss: bool = False, num_threads: int = 0, random_state=None,
                 cg_steps: int = 3, device=None):
        if np.dtype(dtype) != np.float32:
            raise ValueError("only float32 factors are supported")
        self.factors = int(factors)
        self.regularization = float(regularization)
        self.alpha = float(alpha)
        self.dtype = np.dtype(np.float32)
        self.use_native = use_native
        self.use_cg = use_cg
        self.use_gpu = use_gpu
        self.iterations = int(iterations)
        self.calculate_training_loss = calculate_training_loss
        self.num_threads = num_threads
        self.random_state = random_state
        self.cg_steps = int(cg_steps)
        self.device = device
        self.user_factors: Optional[np.ndarray] = None
        self.item_factors: Optional[np.ndarray] = None
        self._reset_caches()

    # ------------------------------------------------------------------ state
    def _reset_caches(self) -> None:
        self._lock = threading.Lock()
        self._dev_items = None       # padded item factors on the device (scoring width)
        self._item_norms = None

    def __getstate__(self):
        return {
            "factors": self.factors, "regularization": self.regularization, "alpha": self.alpha,
            "iterations": self.iterations, "random_state": self.random_state, "cg_steps": self.cg_steps,
            "use_native": self.use_native, "use_cg": self.use_cg, "use_gpu": self.use_gpu,
            "calculate_training_loss": self.calculate_training_loss, "num_threads": self.num_threads,
            "device": None, "dtype": "float32",
            "user_factors": None if self.user_factors is None else np.asarray(self.user_factors, dtype=np.float32),
            "item_factors": None if self.item_factors is None else np.asarray(self.item_factors, dtype=np.float32),
        }

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.dtype = np.dtype(np.float32)
        self._reset_caches()

    # -------------------------------------------------------------------- fit
    def fit(self, user_items, show_progress: bool = True, callback=None) -> None:
        """implicit's ``fit``: cast to fp32, multiply the stored values by
        ``alpha`` when it is not 1, transpose once, initialise X then Y from one
        ``RandomState(random_state)`` stream unless already set, then
        ``iterations`` x {user half-sweep, item half-sweep}."""
        Cui = _check_csr(user_items)
        users, items = Cui.shape
        if Cui.nnz >= 2 ** 31:
            raise ValueError("more than 2^31 stored interactions in one matrix")
        dev = E.require_cuda(self.device)
        rs = self.random_state if isinstance(self.random_state, np.random.RandomState) \
            else np.random.RandomState(self.random_state)
        if self.user_factors is None:
            self.user_factors = rs.rand(users, self.factors).astype(np.float32) * 0.01
        if self.item_factors is None:
            self.item_factors = rs.rand(items, self.factors).astype(np.float32) * 0.01
        if self.user_factors.shape != (users, self.factors) or self.item_factors.shape != (items, self.factors):
            raise ValueError("user_factors / item_factors do not match the matrix shape")

        with self._lock, torch.cuda.device(dev):
            F = E.factor_width(self.factors)
            dCui = E.DeviceCSR.from_scipy(Cui, dev, scale=self.alpha)
            X = E.pad_factors(self.user_factors, F, dev)
            Y = E.pad_factors(self.item_factors, F, dev)
            if self.use_cg:
                eng = E.AlsEngine(dCui, None, X, Y, self.regularization, self.cg_steps)
            else:
                if Cui.nnz and float(np.abs(Cui.data).min()) * abs(self.alpha) < 1.0:
                    raise ValueError("use_cg=False needs |value| * alpha >= 1 for every stored entry "
                                     "(smaller confidences can make a row's system indefinite)")
                # every row with entries goes through the explicit-Gram kernel, which holds the Cholesky solve
                eng = E.AlsEngine(dCui, None, X, Y, self.regularization, self.SOLVE_EXACT, short_max=0)
            for it in range(self.iterations):
                eng.iteration()
                if callback is not None:
                    callback(it, None, None)
            self.user_factors = X[:, :self.factors].contiguous().cpu().numpy()
            self.item_factors = Y[:, :self.factors].contiguous().cpu().numpy()
            self._dev_items = None
            self._item_norms = None

    # -------------------------------------------------------------- inference
    def _device_items(self, dev: torch.device):
        if self._dev_items is None or self._dev_items.device != dev:
            W = E.scoring_width(self.factors, 1)
            self._dev_items = E.pad_factors(self.item_factors, W, dev)
        return self._dev_items

    @property
    def item_norms(self) -> np.ndarray:
        if self._item_norms is None:
            n = np.linalg.norm(self.item_factors, axis=-1).astype(np.float32)
            n[n == 0] = 1e-10
            self._item_norms = n
        return self._item_norms

    def recommend(self, userid, user_items, N: int = 10, filter_already_liked_items: bool = True,
                  filter_items=None, recalculate_user: bool = False, items=None):
        """Top-N items for one user (scalar ``userid`` -> 1-D arrays) or a batch
        (array -> 2-D arrays); ``user_items`` holds one CSR row per user.  Liked
        items and ``filter_items`` are scored -FLT_MAX and, where fewer than N
        other items exist, fill the tail with id -1.  Returns (ids int32,
        scores float32), best first; ties by ascending item index."""
        if self.user_factors is None or self.item_factors is None:
            raise ValueError("model has not been fitted")
        if recalculate_user:
            raise NotImplementedError("recalculate_user=True is not supported")
        if items is not None:
            raise NotImplementedError("restricting the candidate set with items= is not supported")
        scalar = np.isscalar(userid)
        uids = np.atleast_1d(np.asarray(userid)).astype(np.int64)
        mask = None
        if filter_already_liked_items or filter_items is not None:
            if filter_already_liked_items:
                if not sp.isspmatrix_csr(user_items):
                    raise ValueError("user_items needs to be a CSR sparse matrix")
                if user_items.shape[0] != len(uids):
                    raise ValueError("user_items must contain 1 row for every user in userids")
                mask = sp.csr_matrix(user_items)
            else:
                mask = sp.csr_matrix((len(uids), self.item_factors.shape[0]), dtype=np.float32)
            if filter_items is not None:
                fi = np.unique(np.asarray(filter_items, dtype=np.int64))
                extra = sp.csr_matrix((np.ones(len(fi) * len(uids), dtype=np.float32),
                                       (np.repeat(np.arange(len(uids)), len(fi)), np.tile(fi, len(uids)))),
                                      shape=mask.shape)
                mask = sp.csr_matrix(mask + extra)
            if not mask.has_sorted_indices:
                mask = mask.sorted_indices()
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            Yd = self._device_items(dev)
            Q = E.pad_factors(self.user_factors[uids], Yd.shape[1], dev)
            dmask = None if mask is None else E.DeviceCSR.from_scipy(mask, dev)
            ids, scores = E.topk(Q, Yd, int(N), dmask)
            ids = _hide_filtered(ids, scores, dmask)
            ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        if scalar:
            return ids[0], scores[0]
        return ids, scores

    def recommend_all(self, user_items=None, N: int = 10, filter_already_liked_items: bool = True,
                      users: Optional[tuple] = None):
        """The batch form of ``recommend`` for users [u0, u1) (default: all):
        one fused pass, the score matrix never exists."""
        u0, u1 = users if users is not None else (0, self.user_factors.shape[0])
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            Yd = self._device_items(dev)
            Q = E.pad_factors(self.user_factors[u0:u1], Yd.shape[1], dev)
            dmask = None
            if filter_already_liked_items:
                if user_items is None:
                    raise ValueError("user_items is needed to filter liked items")
                m = sp.csr_matrix(user_items)[u0:u1]
                if not m.has_sorted_indices:
                    m = m.sorted_indices()
                dmask = E.DeviceCSR.from_scipy(m, dev)
            ids, scores = E.topk(Q, Yd, int(N), dmask)
            ids = _hide_filtered(ids, scores, dmask)
            return ids.cpu().numpy(), scores.cpu().numpy()

    def similar_items(self, itemid, N: int = 10, recalculate_item: bool = False, item_users=None,
                      filter_items=None, items=None):
        """Cosine-nearest items: scores = (Y . y) / |Y_i| / |y|, best first (the
        item itself comes first, as in implicit)."""
        if recalculate_item or filter_items is not None or items is not None:
            raise NotImplementedError("only the plain similar_items(itemid, N) form is supported")
        scalar = np.isscalar(itemid)
        iids = np.atleast_1d(np.asarray(itemid)).astype(np.int64)
        norms = self.item_norms
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            Yd = self._device_items(dev)
            Q = E.pad_factors(self.item_factors[iids], Yd.shape[1], dev)
            yn = torch.from_numpy(norms).to(dev)
            qn = torch.from_numpy(np.ascontiguousarray(norms[iids])).to(dev)
            ids, scores = E.topk(Q, Yd, int(N), None, y_norm=yn, q_norm=qn)
            ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        if scalar:
            return ids[0], scores[0]
        return ids, scores
