"""LightFM hybrid inference on the GPU.

``LightFMScorer`` stands where a fitted ``lightfm.LightFM`` object stands in the
reference's serving code: it carries the four learned arrays of a LightFM model
(``user_embeddings``, ``user_biases``, ``item_embeddings``, ``item_biases``;
``no_components = 15`` in the reference, ``hybrid_docker/train.py:28-36,157``)
and offers

``predict(user_ids, item_ids, item_features=, user_features=, num_threads=)``
    the call made at ``hybrid_docker/inference.py:159-165,188-194``;
``recommend(user_rows, user_features, item_features, k)``
    predict-all-items + ``scores.argsort()[::-1][:k]`` (``inference.py:197``)
    for any number of users in one fused pass, no liked-item filter.

Representations are feature-embedding segment-sums (identity features included
when the matrices carry them, as ``lightfm.data.Dataset`` builds them); the
score is ``(b_u + b_i) + sum_j u_j i_j`` in that order.  ``LightFM`` adds WARP
training with the adagrad schedule (``fit`` / ``fit_partial``) on the device.
"""

from __future__ import annotations

import ctypes
import threading
from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from . import engine as E


def _identity(n: int) -> sp.csr_matrix:
    return sp.identity(n, dtype=np.float32, format="csr")


class LightFMScorer:
    def __init__(self, user_embeddings, user_biases, item_embeddings, item_biases, user_scale: float = 1.0,
                 item_scale: float = 1.0, device=None):
        self.user_embeddings = np.ascontiguousarray(user_embeddings, dtype=np.float32)
        self.user_biases = np.ascontiguousarray(user_biases, dtype=np.float32)
        self.item_embeddings = np.ascontiguousarray(item_embeddings, dtype=np.float32)
        self.item_biases = np.ascontiguousarray(item_biases, dtype=np.float32)
        if self.user_embeddings.shape[1] != self.item_embeddings.shape[1]:
            raise ValueError("user and item embeddings differ in no_components")
        self.no_components = int(self.user_embeddings.shape[1])
        self.user_scale, self.item_scale = float(user_scale), float(item_scale)
        self.device = device
        self._lock = threading.Lock()
        self._dev = {}
        self._serve = None   # resident state of the per-request path (recommend_request)

    # ------------------------------------------------------------------- state
    # The reference stores and reloads the whole model object (joblib.dump / joblib.load at
    # hybrid_docker/train.py:182 and hybrid_docker/inference.py:102): everything but the lock and the device
    # copies of the tables is plain data.
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items() if k not in ("_lock", "_dev", "_serve")}
        state["device"] = None if self.device is None else str(self.device)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._lock = threading.Lock()
        self._dev = {}
        self._serve = None

    # ----------------------------------------------------------------- helpers
    def _tables(self, dev):
        if self._dev.get("device") != dev:
            self._dev = {
                "device": dev,
                "ue": torch.from_numpy(self.user_embeddings).to(dev), "ub": torch.from_numpy(self.user_biases).to(dev),
                "ie": torch.from_numpy(self.item_embeddings).to(dev), "ib": torch.from_numpy(self.item_biases).to(dev),
            }
        return self._dev

    def _check_features(self, features: Optional[sp.spmatrix], n_rows_default: int, n_feat: int, what: str):
        if features is None:
            features = _identity(n_rows_default)
        features = sp.csr_matrix(features).astype(np.float32, copy=False)
        if features.shape[1] != n_feat:
            raise ValueError(f"The {what} feature matrix specifies {features.shape[1]} features, "
                             f"the model has {n_feat}")
        return features

    def _repr(self, features: sp.csr_matrix, emb: torch.Tensor, bias: torch.Tensor, scale: float, ld: int,
              emb_col: int, bias_col: int, one_col: int, dev) -> torch.Tensor:
        lib = _lib.require()
        csr = E.DeviceCSR.from_scipy(features, dev)
        out = torch.zeros((features.shape[0], ld), dtype=torch.float32, device=dev)
        _lib.check(lib.mirec_lightfm_repr(E._ptr(csr.indptr), E._ptr(csr.indices), E._ptr(csr.data),
                                          features.shape[0], E._ptr(emb), E._ptr(bias), scale, self.no_components,
                                          E._ptr(out), ld, emb_col, bias_col, one_col, E._stream_ptr()),
                   "mirec_lightfm_repr")
        return out

    def representations(self, user_features=None, item_features=None):
        """(user_repr [U, nc+1], item_repr [I, nc+1]) with the bias in the last
        column -- LightFM's ``get_user_representations`` / ``get_item_...``."""
        dev = E.require_cuda(self.device)
        nc = self.no_components
        uf = self._check_features(user_features, self.user_embeddings.shape[0], self.user_embeddings.shape[0], "user")
        itf = self._check_features(item_features, self.item_embeddings.shape[0], self.item_embeddings.shape[0], "item")
        with self._lock, torch.cuda.device(dev):
            t = self._tables(dev)
            ur = self._repr(uf, t["ue"], t["ub"], self.user_scale, nc + 1, 0, nc, -1, dev)
            ir = self._repr(itf, t["ie"], t["ib"], self.item_scale, nc + 1, 0, nc, -1, dev)
            return ur.cpu().numpy(), ir.cpu().numpy()

    # ----------------------------------------------------------------- predict
    def predict(self, user_ids, item_ids, item_features=None, user_features=None, num_threads: int = 1) -> np.ndarray:
        """Scores of explicit (user, item) pairs; a scalar ``user_ids`` is
        repeated for every item, as LightFM does."""
        item_ids = np.ascontiguousarray(item_ids, dtype=np.int32)
        if np.isscalar(user_ids) or np.ndim(user_ids) == 0:
            user_ids = np.full(len(item_ids), int(user_ids), dtype=np.int32)
        user_ids = np.ascontiguousarray(user_ids, dtype=np.int32)
        if len(user_ids) != len(item_ids):
            raise ValueError("user_ids and item_ids differ in length")
        if len(user_ids) and (user_ids.min() < 0 or item_ids.min() < 0):
            raise ValueError("negative ids")
        dev = E.require_cuda(self.device)
        nc = self.no_components
        uf = self._check_features(user_features, self.user_embeddings.shape[0], self.user_embeddings.shape[0], "user")
        itf = self._check_features(item_features, self.item_embeddings.shape[0], self.item_embeddings.shape[0], "item")
        if len(user_ids) and (user_ids.max() >= uf.shape[0] or item_ids.max() >= itf.shape[0]):
            raise ValueError("id exceeds the number of feature rows")
        # only the rows that are asked for are represented
        urows, uinv = np.unique(user_ids, return_inverse=True)
        irows, iinv = np.unique(item_ids, return_inverse=True)
        lib = _lib.require()
        with self._lock, torch.cuda.device(dev):
            t = self._tables(dev)
            ur = self._repr(uf[urows], t["ue"], t["ub"], self.user_scale, nc + 1, 0, nc, -1, dev)
            ir = self._repr(itf[irows], t["ie"], t["ib"], self.item_scale, nc + 1, 0, nc, -1, dev)
            du = torch.from_numpy(uinv.astype(np.int32)).to(dev)
            di = torch.from_numpy(iinv.astype(np.int32)).to(dev)
            out = torch.empty(len(user_ids), dtype=torch.float32, device=dev)
            _lib.check(lib.mirec_lightfm_predict(E._ptr(ur), E._ptr(ir), E._ptr(du), E._ptr(di), len(user_ids), nc,
                                                 E._ptr(out), E._stream_ptr()), "mirec_lightfm_predict")
            return out.cpu().numpy()

    # ------------------------------------------------------------ predict_rank
    def predict_rank(self, test_interactions, train_interactions=None, item_features=None, user_features=None,
                     num_threads: int = 1, check_intersections: bool = True) -> sp.csr_matrix:
        """LightFM's ``predict_rank``: a CSR matrix shaped like
        ``test_interactions`` whose entries are the 0-based ranks of the test
        interactions among the items that are not train positives of the user
        (ties count against the interaction).  Computed on the device by
        ``mirec_lightfm_ranks``; ``predict_ranks`` in lightfm 1.17's
        ``_lightfm_fast`` is the statement it follows."""
        test = sp.csr_matrix(test_interactions)
        test.sort_indices()
        train = None
        if train_interactions is not None:
            train = sp.csr_matrix(train_interactions)
            train.sort_indices()
            if train.shape != test.shape:
                raise ValueError("train and test interactions differ in shape")
            if check_intersections and test.multiply(train).nnz:
                raise ValueError("Test interactions matrix and train interactions matrix share interactions")
        dev = E.require_cuda(self.device)
        nc = self.no_components
        nu, ni = test.shape
        uf = self._check_features(user_features, self.user_embeddings.shape[0], self.user_embeddings.shape[0], "user")
        itf = self._check_features(item_features, self.item_embeddings.shape[0], self.item_embeddings.shape[0], "item")
        if nu > uf.shape[0] or ni > itf.shape[0]:
            raise ValueError("interactions exceed the number of feature rows")
        lib = _lib.require()
        with self._lock, torch.cuda.device(dev):
            t = self._tables(dev)
            ur = self._repr(uf[:nu], t["ue"], t["ub"], self.user_scale, nc + 1, 0, nc, -1, dev)
            ir = self._repr(itf[:ni], t["ie"], t["ib"], self.item_scale, nc + 1, 0, nc, -1, dev)
            tp = torch.from_numpy(test.indptr.astype(np.int64)).to(dev)
            ti = torch.from_numpy(test.indices.astype(np.int32)).to(dev)
            rp = ri = None
            if train is not None:
                rp = torch.from_numpy(train.indptr.astype(np.int64)).to(dev)
                ri = torch.from_numpy(train.indices.astype(np.int32)).to(dev)
            ranks = torch.zeros(max(test.nnz, 1), dtype=torch.float32, device=dev)
            _lib.check(lib.mirec_lightfm_ranks(E._ptr(ur), E._ptr(ir), nu, ni, nc, E._ptr(tp), E._ptr(ti), E._ptr(rp),
                                               E._ptr(ri), E._ptr(ranks), E._stream_ptr()), "mirec_lightfm_ranks")
            data = ranks[:test.nnz].cpu().numpy()
        return sp.csr_matrix((data, test.indices.copy(), test.indptr.copy()), shape=test.shape)

    # --------------------------------------------------------------- recommend
    def recommend(self, user_rows, user_features, item_features=None, k: int = 20):
        """Top-k items for each listed row of ``user_features`` (None = all
        rows).  Returns (ids int32 [n, k], scores fp32 [n, k]), best first, ties
        by ascending item index."""
        dev = E.require_cuda(self.device)
        nc = self.no_components
        uf = self._check_features(user_features, self.user_embeddings.shape[0], self.user_embeddings.shape[0], "user")
        itf = self._check_features(item_features, self.item_embeddings.shape[0], self.item_embeddings.shape[0], "item")
        if user_rows is not None:
            uf = uf[np.atleast_1d(np.asarray(user_rows, dtype=np.int64))]
        W = E.scoring_width(nc + 2, int(k))
        with self._lock, torch.cuda.device(dev):
            t = self._tables(dev)
            # operand rows [b_u, 1, e...] and [1, b_i, e...]: the MFMA chain then adds b_u, b_i, u_0 i_0, ...
            Q = self._repr(uf, t["ue"], t["ub"], self.user_scale, W, 2, 0, 1, dev)
            Y = self._repr(itf, t["ie"], t["ib"], self.item_scale, W, 2, 1, 0, dev)
            n = Q.shape[0]
            if n <= (1 << 17):
                ids, scores = E.topk(Q, Y, int(k))
                return ids.cpu().numpy(), scores.cpu().numpy()
            # many users: chunk by chunk, each chunk's download overlapping the next chunk's kernels
            dl = E.ChunkedDownload(n, int(k), dev)
            order = E.norm_order(Y)       # one sweep order for all chunks
            for a, b in dl.ranges():
                ids, scores = E.topk(Q[a:b], Y, int(k), path="sweep", item_order=order)
                dl.submit(ids, scores, a)
            return dl.finish()


    # ---------------------------------------------------------- request path
    def _serving_state(self, dev, user_features, item_features, k: int):
        """What a request needs, resident on the device while the caller keeps passing the same feature matrices
        (the reference's ``state`` dict does, hybrid_docker/inference.py:116-123): the item representations as
        scoring rows, the user feature matrix, one query row, and output / download / workspace buffers."""
        nc = self.no_components
        key = (str(dev), id(user_features), getattr(user_features, "shape", None), getattr(user_features, "nnz", None),
               id(item_features), getattr(item_features, "shape", None), getattr(item_features, "nnz", None), int(k))
        st = self._serve
        if st is not None and st["key"] == key:
            return st
        lib = _lib.require()
        uf = self._check_features(user_features, self.user_embeddings.shape[0], self.user_embeddings.shape[0], "user")
        itf = self._check_features(item_features, self.item_embeddings.shape[0], self.item_embeddings.shape[0], "item")
        W = E.factor_width(nc + 2, for_scoring=True)
        t = self._tables(dev)
        ni = itf.shape[0]
        st = {
            "key": key, "hold": (user_features, item_features), "W": W, "n_users": uf.shape[0], "n_feat": uf.shape[1],
            "Y": self._repr(itf, t["ie"], t["ib"], self.item_scale, W, 2, 1, 0, dev),
            "uf": E.DeviceCSR.from_scipy(uf, dev),
            "Q": torch.zeros((1, W), dtype=torch.float32, device=dev),
            "out": torch.empty(2 * k, dtype=torch.int32, device=dev),
            "host": torch.empty(2 * k, dtype=torch.int32).pin_memory(),
            "ws": torch.empty(int(lib.mirec_topk_small_workspace_bytes(1, ni, int(k))), dtype=torch.uint8, device=dev),
        }
        self._serve = st
        return st

    def recommend_request(self, user_row, user_features, item_features=None, k: int = 20, feature_row=None):
        """Top-k items for ONE user, the form the reference serves (hybrid_docker/inference.py:156-165,197: one
        ``predict`` over all items and an argsort per request): ``user_row`` names a row of ``user_features``; for
        a user without one, ``feature_row`` is its 1 x n_features CSR row (the cold-start row of
        inference.py:66-83).  The item representations and the feature matrix stay on the device between
        requests; a known user's request uploads nothing.  Returns (ids int32 [k], scores fp32 [k])."""
        dev = E.require_cuda(self.device)
        lib = _lib.require()
        k = int(k)
        nc = self.no_components
        with self._lock, torch.cuda.device(dev):
            st = self._serving_state(dev, user_features, item_features, k)
            t = self._tables(dev)
            keep = None
            if feature_row is None:
                u = int(user_row)
                if not 0 <= u < st["n_users"]:
                    raise IndexError(f"user row {u} is out of range")
                d = st["uf"]
                ptrs = (d.indptr.data_ptr() + 8 * u, d.indices.data_ptr(), d.data.data_ptr())
            else:
                row = sp.csr_matrix(feature_row, dtype=np.float32)
                if row.shape != (1, st["n_feat"]):
                    raise ValueError(f"feature_row must be 1 x {st['n_feat']}")
                n = int(row.nnz)
                if n and (row.indices.min() < 0 or row.indices.max() >= st["n_feat"]):
                    raise ValueError("feature_row names a feature the model does not have")
                keep = (torch.tensor([0, n], dtype=torch.int64, device=dev),
                        torch.from_numpy(np.ascontiguousarray(row.indices if n else np.zeros(1), dtype=np.int32)).to(dev),
                        torch.from_numpy(np.ascontiguousarray(row.data if n else np.zeros(1), dtype=np.float32)).to(dev))
                ptrs = tuple(x.data_ptr() for x in keep)
            stream = torch.cuda.current_stream().cuda_stream
            W, Q, out = st["W"], st["Q"], st["out"]
            _lib.check(lib.mirec_lightfm_repr(ptrs[0], ptrs[1], ptrs[2], 1, E._ptr(t["ue"]), E._ptr(t["ub"]),
                                              self.user_scale, nc, Q.data_ptr(), W, 2, 0, 1, stream), "mirec_lightfm_repr")
            base = out.data_ptr()
            _lib.check(lib.mirec_topk_small(Q.data_ptr(), 1, st["Y"].data_ptr(), st["Y"].shape[0], W, 0, 0, 0, 0, k, base,
                                            base + 4 * k, st["ws"].data_ptr(), st["ws"].numel(), stream),
                       "mirec_topk_small")
            st["host"].copy_(out, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            arr = st["host"].numpy()
            del keep
            return arr[:k].copy(), arr[k:].view(np.float32).copy()


class LightFM(LightFMScorer):
    """The part of ``lightfm.LightFM`` the reference trains with
    (``hybrid_docker/train.py:157``: ``LightFM(no_components=, loss='warp',
    random_state=, learning_schedule='adagrad', item_alpha=1e-6)`` and
    ``fit_partial(train, user_features=, item_features=, epochs=1,
    num_threads=4)`` at ``hybrid_docker/utils.py:100-107``), on the device.
    Everything a fitted model is asked for afterwards (``predict``,
    ``predict_rank``, ``recommend``) is inherited from the scorer.

    Examples of an epoch run concurrently and update shared parameters without
    locks -- the library's Hogwild contract -- so two fits differ in the last
    digits and parity with a sequential run is statistical; ``serial=True``
    runs the examples one after the other and is reproducible bit for bit."""

    def __init__(self, no_components: int = 10, k: int = 5, n: int = 10, learning_schedule: str = "adagrad",
                 loss: str = "logistic", learning_rate: float = 0.05, rho: float = 0.95, epsilon: float = 1e-6,
                 item_alpha: float = 0.0, user_alpha: float = 0.0, max_sampled: int = 10, random_state=None,
                 device=None, serial: bool = False):
        if loss != "warp":
            raise NotImplementedError(f"loss={loss!r}: only 'warp' (the reference's setting) is built")
        if learning_schedule != "adagrad":
            raise NotImplementedError(f"learning_schedule={learning_schedule!r}: only 'adagrad' is built")
        if not (item_alpha >= 0.0 and user_alpha >= 0.0 and no_components > 0 and max_sampled > 0):
            raise ValueError("no_components and max_sampled must be positive, the alphas non-negative")
        nc = int(no_components)
        empty = np.zeros((0, nc), dtype=np.float32)
        super().__init__(empty, np.zeros(0, np.float32), empty, np.zeros(0, np.float32), device=device)
        self.loss, self.learning_schedule = loss, learning_schedule
        self.learning_rate, self.rho, self.epsilon = float(learning_rate), float(rho), float(epsilon)
        self.k, self.n = int(k), int(n)
        self.item_alpha, self.user_alpha, self.max_sampled = float(item_alpha), float(user_alpha), int(max_sampled)
        if random_state is None:
            self.random_state = np.random.RandomState()
        elif isinstance(random_state, np.random.RandomState):
            self.random_state = random_state
        else:
            self.random_state = np.random.RandomState(random_state)
        self.serial = bool(serial)
        self.item_embedding_gradients = self.item_bias_gradients = None
        self.user_embedding_gradients = self.user_bias_gradients = None
        self._examples_seen = 0

    # The eight parameter tables live on the device between ``fit_partial`` calls (the reference's training loop
    # calls it once per epoch, hybrid_docker/utils.py:100-107): an epoch neither uploads nor downloads them.  The
    # numpy attributes of the same names are read through properties that fetch a table when it is asked for
    # after the device has moved on; assigning one of them drops the device copies.
    _PARAMS = ("item_embeddings", "item_embedding_gradients", "item_biases", "item_bias_gradients",
               "user_embeddings", "user_embedding_gradients", "user_biases", "user_bias_gradients")

    def _sync_params_to_host(self) -> None:
        st = self.__dict__.get("_dev_par")
        if st is not None:
            for name in sorted(st["stale"]):
                self.__dict__["_h_" + name] = st["t"][name].cpu().numpy()
            st["stale"].clear()

    def _device_params(self, dev):
        """The tables as device tensors, uploaded only when the host copies are newer."""
        st = self.__dict__.get("_dev_par")
        if st is None or st["device"] != dev:
            self._sync_params_to_host()
            def up(a):
                a = np.ascontiguousarray(a, dtype=np.float32)
                return torch.from_numpy(a if a.flags.writeable else a.copy()).to(dev)   # (snapshots are read-only)

            st = {"device": dev, "stale": set(), "t": {n: up(self.__dict__["_h_" + n]) for n in self._PARAMS}}
            self.__dict__["_dev_par"] = st
        return st

    def _tables(self, dev):
        st = self.__dict__.get("_dev_par")
        if st is not None and st["device"] == dev:   # score from the tables the training updates in place
            t = st["t"]
            return {"device": dev, "ue": t["user_embeddings"], "ub": t["user_biases"], "ie": t["item_embeddings"],
                    "ib": t["item_biases"]}
        return super()._tables(dev)

    def __getstate__(self):
        self._sync_params_to_host()
        state = {k: v for k, v in super().__getstate__().items() if k not in ("_dev_par", "_train_inputs")}
        for n in self._PARAMS:
            state[n] = state.pop("_h_" + n, None)
        return state

    def __setstate__(self, state):
        state = dict(state)
        for n in self._PARAMS:
            if n in state:
                state["_h_" + n] = state.pop(n)
        super().__setstate__(state)
        self.__dict__["_dev_par"] = None
        self.__dict__["_train_inputs"] = None

    def _initialize(self, no_user_features: int, no_item_features: int) -> None:
        # lightfm.LightFM._initialize: items first, then users; adagrad accumulators start at one
        nc = self.no_components
        rs = self.random_state
        self.item_embeddings = ((rs.rand(no_item_features, nc) - 0.5) / nc).astype(np.float32)
        self.item_embedding_gradients = np.ones((no_item_features, nc), dtype=np.float32)
        self.item_biases = np.zeros(no_item_features, dtype=np.float32)
        self.item_bias_gradients = np.ones(no_item_features, dtype=np.float32)
        self.user_embeddings = ((rs.rand(no_user_features, nc) - 0.5) / nc).astype(np.float32)
        self.user_embedding_gradients = np.ones((no_user_features, nc), dtype=np.float32)
        self.user_biases = np.zeros(no_user_features, dtype=np.float32)
        self.user_bias_gradients = np.ones(no_user_features, dtype=np.float32)
        self._examples_seen = 0   # what the accumulators have absorbed: sets how many examples may run side by side

    def fit(self, interactions, user_features=None, item_features=None, sample_weight=None, epochs: int = 1,
            num_threads: int = 1, verbose: bool = False):
        self.item_embedding_gradients = None  # start afresh
        return self.fit_partial(interactions, user_features, item_features, sample_weight, epochs, num_threads, verbose)

    @staticmethod
    def _fingerprint(m):
        """Identity of an input matrix between calls: the object itself plus a digest of ALL of its arrays (a
        caller may refill the same object in place, with the same number of entries)."""
        if m is None:
            return None
        import xxhash
        h = xxhash.xxh64()
        for name in ("data", "indices", "indptr", "row", "col"):
            a = getattr(m, name, None)
            if a is not None:
                h.update(np.ascontiguousarray(a).view(np.uint8))
        return (id(m), type(m).__name__, m.shape, int(m.nnz), h.intdigest())

    def _device_inputs(self, dev, interactions, user_features, item_features, sample_weight):
        """Interactions, feature matrices and sample weights as device arrays; kept from call to call while the
        caller passes the same objects (the reference's epoch loop does)."""
        key = (str(dev),) + tuple(self._fingerprint(m) for m in (interactions, user_features, item_features, sample_weight))
        cached = self.__dict__.get("_train_inputs")
        if cached is not None and cached["key"] == key:
            return cached
        coo = sp.coo_matrix(interactions)
        if coo.dtype != np.float32:
            coo = coo.astype(np.float32)
        n_users, n_items = coo.shape
        uf = sp.csr_matrix(_identity(n_users) if user_features is None else user_features, dtype=np.float32)
        itf = sp.csr_matrix(_identity(n_items) if item_features is None else item_features, dtype=np.float32)
        if uf.shape[0] < n_users or itf.shape[0] < n_items:
            raise Exception("Number of feature rows does not cover the interactions")
        sw = None
        if sample_weight is not None:
            swm = sp.coo_matrix(sample_weight)
            if swm.shape != coo.shape or not (np.array_equal(swm.row, coo.row) and np.array_equal(swm.col, coo.col)):
                raise ValueError("Sample weight and interaction matrix entries must be in the same order")
            sw = np.ascontiguousarray(swm.data, dtype=np.float32)
        pos = sp.csr_matrix(coo)
        pos.sort_indices()
        up = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).to(dev)
        cached = {
            "key": key, "hold": (interactions, user_features, item_features, sample_weight),  # keeps the ids alive
            "n_users": n_users, "n_items": n_items, "n_uf": uf.shape[1], "n_if": itf.shape[1], "nnz": int(coo.nnz),
            "ifp": up(itf.indptr, np.int64), "ifi": up(itf.indices, np.int32), "ifd": up(itf.data, np.float32),
            "ufp": up(uf.indptr, np.int64), "ufi": up(uf.indices, np.int32), "ufd": up(uf.data, np.float32),
            "pp": up(pos.indptr, np.int64), "pi": up(pos.indices, np.int32),
            "uid": up(coo.row, np.int32), "iid": up(coo.col, np.int32), "y": up(coo.data, np.float32),
            "sw": None if sw is None else up(sw, np.float32),
        }
        self.__dict__["_train_inputs"] = cached
        return cached

    def fit_partial(self, interactions, user_features=None, item_features=None, sample_weight=None, epochs: int = 1,
                    num_threads: int = 1, verbose: bool = False):
        if num_threads < 1:
            raise ValueError("Number of threads must be 1 or larger.")
        dev = E.require_cuda(self.device)
        lib = _lib.require()
        nc = self.no_components
        with self._lock, torch.cuda.device(dev):
            d = self._device_inputs(dev, interactions, user_features, item_features, sample_weight)
            # (looked up without the property getters: those download a table the device holds the newer copy of)
            dp = self.__dict__.get("_dev_par")

            def rows_of(name):
                if dp is not None:
                    return int(dp["t"][name].shape[0])
                h = self.__dict__.get("_h_" + name)
                return None if h is None else int(h.shape[0])

            if rows_of("item_embedding_gradients") is None:
                self._initialize(d["n_uf"], d["n_if"])
            if d["n_uf"] != rows_of("user_embeddings") or d["n_if"] != rows_of("item_embeddings"):
                raise ValueError("Incorrect number of features in user_features / item_features")
            st = self._device_params(dev)
            par = st["t"]
            ws_bytes = int(lib.mirec_lightfm_warp_workspace_bytes(self.max_sampled))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            for _ in range(int(epochs)):
                # one permutation and one generator seed per epoch from the model's random state, as the library draws them
                order = np.arange(d["nnz"], dtype=np.int32)
                self.random_state.shuffle(order)
                seed = int(self.random_state.randint(0, np.iinfo(np.int32).max))
                d_order = torch.from_numpy(order).to(dev)
                st["stale"].update(self._PARAMS)   # from here on the device holds the newer tables
                _lib.check(lib.mirec_lightfm_warp_epoch(
                    E._ptr(d["ifp"]), E._ptr(d["ifi"]), E._ptr(d["ifd"]), d["n_items"], d["n_if"],
                    E._ptr(d["ufp"]), E._ptr(d["ufi"]), E._ptr(d["ufd"]), d["n_uf"],
                    E._ptr(d["pp"]), E._ptr(d["pi"]), E._ptr(d["uid"]), E._ptr(d["iid"]), E._ptr(d["y"]), E._ptr(d["sw"]),
                    E._ptr(d_order), d["nnz"],
                    E._ptr(par["item_embeddings"]), E._ptr(par["item_embedding_gradients"]), E._ptr(par["item_biases"]),
                    E._ptr(par["item_bias_gradients"]), E._ptr(par["user_embeddings"]),
                    E._ptr(par["user_embedding_gradients"]), E._ptr(par["user_biases"]), E._ptr(par["user_bias_gradients"]),
                    nc, self.learning_rate, self.item_alpha, self.user_alpha, self.max_sampled, seed,
                    1 if self.serial else 0, self._examples_seen, E._ptr(ws), ws_bytes, E._stream_ptr()),
                    "mirec_lightfm_warp_epoch")
                self._examples_seen += d["nnz"]
            self._dev = {}  # device tables uploaded from older host copies are stale
            self._serve = None
            finite = all(bool(torch.isfinite(par[n]).all().item()) for n in ("item_embeddings", "user_embeddings", "item_biases"))
        if not finite:
            raise ValueError("Not all estimated parameters are finite, your model may have diverged. "
                             "Try decreasing the learning rate or normalising feature values and sample weights")
        return self


def _param_property(name: str) -> property:
    """A parameter table as a numpy array.  While the device holds the newer copy (after ``fit_partial``) the getter
    downloads it once and hands out that SNAPSHOT, marked read-only: writing into it would not reach the model --
    assign a whole array to the attribute instead (the setter makes the host copies lead again)."""
    def getter(self):
        st = self.__dict__.get("_dev_par")
        if st is not None and name in st["stale"]:
            snap = st["t"][name].cpu().numpy()
            snap.setflags(write=False)
            self.__dict__["_h_" + name] = snap
            st["stale"].discard(name)
        return self.__dict__.get("_h_" + name)

    def setter(self, value):
        if self.__dict__.get("_dev_par") is not None:   # a table written by the caller: the host copies lead again
            self._sync_params_to_host()
            self.__dict__["_dev_par"] = None
        self.__dict__["_serve"] = None
        self.__dict__["_h_" + name] = value

    return property(getter, setter)


for _name in LightFM._PARAMS:
    setattr(LightFM, _name, _param_property(_name))
del _name


# The cold-start vocabulary of the reference's request path (hybrid_docker/inference.py:168-177), in its order:
# one token of each list is drawn for a user the model has never seen.
COLD_START_GENDERS = ("M", "F")
COLD_START_JOBS = (
    "other or not specified", "executive/managerial", "sales/marketing", "college/grad student",
    "doctor/health care", "academic/educator", "homemaker", "K-12 student", "self-employed", "scientist",
    "technician/engineer", "clerical/admin", "artist", "tradesman/craftsman", "retired", "unemployed",
    "programmer", "customer service", "writer", "lawyer", "farmer",
)
COLD_START_AGES = ("25-34", "18-24", "35-44", "55-64", "45-54", "65+", "0-17")


def draw_cold_start_tokens(rng=None):
    """The three ``age:/job:/gender:`` tokens of an unknown user, drawn in the
    reference's order (age, job, gender) with ``rng.choice`` -- ``rng`` is the
    ``random`` module by default, the generator the reference draws from, or any
    ``random.Random`` (seeded, a test reproduces the draw)."""
    import random as _random
    rng = _random if rng is None else rng
    return [f"age:{rng.choice(COLD_START_AGES)}", f"job:{rng.choice(COLD_START_JOBS)}",
            f"gender:{rng.choice(COLD_START_GENDERS)}"]


def _user_feature_mapping(state) -> dict:
    """token -> feature column: ``state["user_feature_mapping"]`` when given,
    else the private map of the pickled ``lightfm.data.Dataset`` the reference
    reads (``dataset._user_feature_mapping``, hybrid_docker/inference.py:72)."""
    fmap = state.get("user_feature_mapping")
    if fmap is None and state.get("dataset") is not None:
        fmap = getattr(state["dataset"], "_user_feature_mapping", None)
    return {} if fmap is None else fmap


def build_coldstart_user_row(fmap: dict, tokens, n_features: Optional[int] = None) -> sp.csr_matrix:
    """1 x n_features CSR row with a one in the column of every known token
    (hybrid_docker/inference.py:66-83); tokens the vocabulary lacks are skipped,
    no match at all leaves a zero row (the scores are then the item biases)."""
    cols = [fmap[t] for t in tokens if t in fmap]
    n = len(fmap) if n_features is None else int(n_features)
    return sp.csr_matrix((np.ones(len(cols), dtype=np.float32), ([0] * len(cols), cols)), shape=(1, n))


def compute_recommendations(user_id, state, top_k: int = 20, num_threads: int = 4, cold_start_tokens=None, rng=None):
    """``compute_recommendations`` of the reference's request path
    (``hybrid_docker/inference.py:129-200``; called as
    ``compute_recommendations(user_id, state, top_k=top_k)`` at ``:238``) on the
    GPU scorer: same positional arguments, same result (external item ids, best
    first).

    ``state`` is what ``load_state`` (``inference.py:89-123``) returns:
    ``model`` (a LightFMScorer), ``user_features``, ``item_features``,
    ``inv_item_map``, ``user_to_idx`` and, for unknown users, the token ->
    column map (``user_feature_mapping`` or ``dataset._user_feature_mapping``).
    A known user is scored from its row of ``user_features``; an unknown one
    from three tokens drawn at random from the reference's vocabulary
    (``draw_cold_start_tokens``; ``rng`` seeds the draw, ``cold_start_tokens``
    replaces it).  ``num_threads`` is accepted and ignored: there are no host
    threads to size.  Equal scores come out in ascending item order, where
    numpy's ``argsort()[::-1]`` leaves their order unspecified."""
    del num_threads
    model: LightFMScorer = state["model"]
    if user_id in state["user_to_idx"]:
        ids, _ = model.recommend_request(state["user_to_idx"][user_id], state["user_features"], state["item_features"],
                                         top_k)
    else:
        tokens = draw_cold_start_tokens(rng) if cold_start_tokens is None else list(cold_start_tokens)
        row = build_coldstart_user_row(_user_feature_mapping(state), tokens, state["user_features"].shape[1])
        ids, _ = model.recommend_request(None, state["user_features"], state["item_features"], top_k, feature_row=row)
    return [state["inv_item_map"][int(i)] for i in ids if i >= 0]


# ---------------------------------------------------------------------------
# lightfm.evaluation, as called at hybrid_docker/utils.py:111-155
def _rows_with_interactions(test: sp.csr_matrix, values: np.ndarray, preserve_rows: bool) -> np.ndarray:
    return values if preserve_rows else values[np.diff(test.indptr) > 0]


def precision_at_k(model: LightFMScorer, test_interactions, train_interactions=None, k: int = 10, user_features=None,
                   item_features=None, preserve_rows: bool = False, num_threads: int = 1,
                   check_intersections: bool = True) -> np.ndarray:
    """Per-user fraction of the top k that are test positives (``lightfm.evaluation.precision_at_k``)."""
    ranks = model.predict_rank(test_interactions, train_interactions, item_features, user_features, num_threads,
                               check_intersections)
    ranks.data = np.less(ranks.data, k, ranks.data)
    precision = np.squeeze(np.array(ranks.sum(axis=1)), axis=1) / k
    return _rows_with_interactions(sp.csr_matrix(test_interactions), precision, preserve_rows)


def recall_at_k(model: LightFMScorer, test_interactions, train_interactions=None, k: int = 10, user_features=None,
                item_features=None, preserve_rows: bool = False, num_threads: int = 1,
                check_intersections: bool = True) -> np.ndarray:
    """Per-user fraction of the test positives that are in the top k (``lightfm.evaluation.recall_at_k``)."""
    ranks = model.predict_rank(test_interactions, train_interactions, item_features, user_features, num_threads,
                               check_intersections)
    ranks.data = np.less(ranks.data, k, ranks.data)
    test = sp.csr_matrix(test_interactions)
    retrieved = np.squeeze(np.array(test.getnnz(axis=1)))
    hit = np.squeeze(np.array(ranks.sum(axis=1)), axis=1)
    if not preserve_rows:
        keep = np.diff(test.indptr) > 0
        return hit[keep] / retrieved[keep]
    return hit / np.maximum(retrieved, 1)


def auc_from_ranks(ranks: sp.csr_matrix, num_train_positives: np.ndarray) -> np.ndarray:
    """lightfm's ``calculate_auc_from_rank``: per user, the probability that a
    random test positive outranks a random negative; 0.5 where a class is
    missing.  The terms are formed in float32 as in the library and summed in
    float64 (the library's float32 running sum differs by rounding only)."""
    n_users, n_items = ranks.shape
    indptr = ranks.indptr.astype(np.int64)
    num_pos = np.diff(indptr)
    num_neg = n_items - (num_pos + num_train_positives.astype(np.int64))
    rows = np.repeat(np.arange(n_users, dtype=np.int64), num_pos)
    order = np.lexsort((ranks.data, rows))
    sorted_ranks = ranks.data[order].astype(np.float32)
    within = (np.arange(len(rows), dtype=np.int64) - indptr[rows]).astype(np.float32)
    adj = np.maximum(sorted_ranks - within, np.float32(0))
    with np.errstate(divide="ignore", invalid="ignore"):
        terms = (np.float32(1) - adj / num_neg[rows].astype(np.float32)).astype(np.float32)
    sums = np.zeros(n_users, dtype=np.float64)
    np.add.at(sums, rows, terms.astype(np.float64))
    auc = np.full(n_users, 0.5, dtype=np.float32)
    ok = (num_pos > 0) & (num_neg != n_items)
    auc[ok] = (sums[ok] / num_pos[ok]).astype(np.float32)
    return auc


def auc_score(model: LightFMScorer, test_interactions, train_interactions=None, user_features=None,
              item_features=None, preserve_rows: bool = False, num_threads: int = 1,
              check_intersections: bool = True) -> np.ndarray:
    """``lightfm.evaluation.auc_score`` (hybrid_docker/utils.py:111-117,135-140)."""
    ranks = model.predict_rank(test_interactions, train_interactions, item_features, user_features, num_threads,
                               check_intersections)
    test = sp.csr_matrix(test_interactions)
    ntp = (np.zeros(test.shape[0], dtype=np.int32) if train_interactions is None
           else np.asarray(sp.csr_matrix(train_interactions).getnnz(axis=1), dtype=np.int32))
    return _rows_with_interactions(test, auc_from_ranks(ranks, ntp), preserve_rows)
