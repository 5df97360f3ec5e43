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
score is ``(b_u + b_i) + sum_j u_j i_j`` in that order.  Training (WARP) is out
of scope; arrays come from a trained model or, in benchmarks, from a generator.
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
        features = sp.csr_matrix(features).astype(np.float32)
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
            ids, scores = E.topk(Q, Y, int(k))
            return ids.cpu().numpy(), scores.cpu().numpy()


def compute_recommendations(user_id, state, top_k: int = 20, cold_start_tokens=None):
    """The request path of ``compute_recommendations``
    (``hybrid_docker/inference.py:129-200``) on the GPU scorer.  ``state`` holds
    ``model`` (a LightFMScorer), ``user_features``, ``item_features``,
    ``inv_item_map``, ``user_to_idx`` and, for unknown users,
    ``user_feature_mapping`` (token -> feature column); ``cold_start_tokens``
    are the three ``age:/job:/gender:`` tokens the reference draws at random."""
    model: LightFMScorer = state["model"]
    if user_id in state["user_to_idx"]:
        ids, _ = model.recommend([state["user_to_idx"][user_id]], state["user_features"], state["item_features"], top_k)
    else:
        fmap = state.get("user_feature_mapping", {})
        cols = [fmap[t] for t in (cold_start_tokens or []) if t in fmap]
        n_feat = state["user_features"].shape[1]
        row = sp.csr_matrix((np.ones(len(cols), dtype=np.float32), ([0] * len(cols), cols)), shape=(1, n_feat))
        ids, _ = model.recommend(None, row, state["item_features"], top_k)
    return [state["inv_item_map"][int(i)] for i in ids[0] if i >= 0]
