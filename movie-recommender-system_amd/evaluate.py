"""Held-out ranking evaluation on the device.

The reference evaluates its ALS model in notebooks/baseline_als.ipynb (cells
66ca94ea, fd92678d): per user with held-out items, hits = |recs[:k] n actual|,
precision = hits / len(recs), recall = hits / len(actual), F1 their harmonic
mean, hit-rate = any hit, all averaged over those users.  LightFM's
``precision_at_k`` / ``recall_at_k`` (called at hybrid_docker/utils.py:111-155)
divide by k instead of len(recs).  Both are computed here from one kernel pass
over the top-k ids (``mirec_ranking_hits``); the per-user counts are integers and
the averages are accumulated on the host in user order, so the result equals the
CPU evaluation bit for bit.  LightFM's ``auc_score`` needs the rank of every
held-out item among all items: that is ``lightfm.auc_score`` (device ranks
from ``mirec_lightfm_ranks``).
"""

from __future__ import annotations

from typing import Dict

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .engine import DeviceCSR, _ptr, _stream_ptr


def ranking_hits(rec_ids: torch.Tensor, held_out: DeviceCSR):
    """(hits, n_rec) int32 device tensors, one entry per user."""
    lib = _lib.require()
    assert rec_ids.dtype == torch.int32 and rec_ids.is_contiguous() and rec_ids.dim() == 2
    nu, k = rec_ids.shape
    assert held_out.shape[0] == nu, "one held-out row per user"
    hits = torch.empty(nu, dtype=torch.int32, device=rec_ids.device)
    nrec = torch.empty(nu, dtype=torch.int32, device=rec_ids.device)
    _lib.check(lib.mirec_ranking_hits(_ptr(rec_ids), nu, k, _ptr(held_out.indptr), _ptr(held_out.indices),
                                      _ptr(hits), _ptr(nrec), _stream_ptr()), "mirec_ranking_hits")
    return hits, nrec


def _sequential_mean(values: np.ndarray, n: int) -> float:
    # the CPU evaluation adds user by user; cumsum keeps that order (np.sum would add pairwise)
    return float(np.cumsum(values, dtype=np.float64)[-1] / n) if values.size else 0.0


def ranking_metrics(rec_ids, held_out, k: int = 20, device=None) -> Dict[str, float]:
    """precision / recall / F1 / hit-rate @k over users that have held-out items
    (the notebook's definition) plus ``precision_at_k`` / ``recall_at_k`` in
    LightFM's definition (hits / k)."""
    if isinstance(held_out, DeviceCSR):
        dev = held_out.indptr.device
        dcsr = held_out
    else:
        M = sp.csr_matrix(held_out)
        M.sort_indices()
        dev = torch.device(device) if device is not None else (
            rec_ids.device if isinstance(rec_ids, torch.Tensor) else torch.device("cuda"))
        dcsr = DeviceCSR.from_scipy(M, dev)
    if not isinstance(rec_ids, torch.Tensor):
        rec_ids = torch.from_numpy(np.ascontiguousarray(rec_ids, dtype=np.int32))
    rec = rec_ids.to(dev)[:, :k].contiguous()
    hits_d, nrec_d = ranking_hits(rec, dcsr)
    hits = hits_d.cpu().numpy().astype(np.float64)
    nrec = nrec_d.cpu().numpy().astype(np.float64)
    indptr = dcsr.indptr.cpu().numpy()
    nact = (indptr[1:] - indptr[:-1]).astype(np.float64)
    sel = nact > 0
    hits, nrec, nact = hits[sel], nrec[sel], nact[sel]
    n = max(int(sel.sum()), 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(nrec > 0, hits / nrec, 0.0)
        r = hits / nact
        f = np.where(p + r > 0, 2 * p * r / (p + r), 0.0)
    return {"precision": _sequential_mean(p, n), "recall": _sequential_mean(r, n), "f1": _sequential_mean(f, n),
            "hit_rate": _sequential_mean((hits > 0).astype(np.float64), n), "users": int(sel.sum()) or 1,
            "precision_at_k": _sequential_mean(hits / float(k), n), "recall_at_k": _sequential_mean(r, n)}
