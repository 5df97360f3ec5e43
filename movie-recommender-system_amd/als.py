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

import os
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


class _ServingState:
    """Device-resident state behind the per-request calls: nothing here is uploaded again per request."""

    def __init__(self, dev: torch.device, Y: torch.Tensor, X: torch.Tensor, norms: torch.Tensor):
        self.dev, self.Y, self.X, self.norms = dev, Y, X, norms
        self._bufs = {}

    def buffers(self, N: int):
        """(device int32[2N]: ids then score bits, its pinned host twin, the workspace of one query) for lists of
        length N."""
        b = self._bufs.get(N)
        if b is None:
            lib = _lib.require()
            out = torch.empty(2 * N, dtype=torch.int32, device=self.dev)
            host = torch.empty(2 * N, dtype=torch.int32).pin_memory()
            ws = torch.empty(int(lib.mirec_topk_small_workspace_bytes(1, self.Y.shape[0], N)), dtype=torch.uint8,
                             device=self.dev)
            b = self._bufs[N] = (out, host, ws)
        return b


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
                 cg_steps: int = 3, device=None, devices=None):
        """``device``: the HIP device of a single-GPU fit and of scoring (default: the current one).
        ``devices``: a list of device indices, one rank each, for a fit sharded over the GPUs of one node
        (``fit`` then runs in child processes, see ``_fit_sharded``); scoring uses ``device`` or the first of
        them."""
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
        self.devices = None if devices is None else [int(d) for d in devices]
        if self.devices is not None and not self.devices:
            raise ValueError("devices must name at least one device")
        self.device = device if device is not None or not self.devices else f"cuda:{self.devices[0]}"
        self.user_factors: Optional[np.ndarray] = None
        self.item_factors: Optional[np.ndarray] = None
        self._reset_caches()

    # ------------------------------------------------------------------ state
    def _reset_caches(self) -> None:
        self._lock = threading.Lock()
        self._dev_items = None       # padded item factors on the device (scoring width)
        self._item_norms = None
        self._serving = None         # device-resident state of the per-request calls (_ServingState)
        self._attached = None        # (host CSR, DeviceCSR) of attach_user_items

    def __getstate__(self):
        return {
            "factors": self.factors, "regularization": self.regularization, "alpha": self.alpha,
            "iterations": self.iterations, "random_state": self.random_state, "cg_steps": self.cg_steps,
            "use_native": self.use_native, "use_cg": self.use_cg, "use_gpu": self.use_gpu,
            "calculate_training_loss": self.calculate_training_loss, "num_threads": self.num_threads,
            "device": None, "devices": self.devices, "dtype": "float32",
            "user_factors": None if self.user_factors is None else np.asarray(self.user_factors, dtype=np.float32),
            "item_factors": None if self.item_factors is None else np.asarray(self.item_factors, dtype=np.float32),
        }

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.setdefault("devices", None)
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
        if not self.use_cg and Cui.nnz and float(np.abs(Cui.data).min()) * abs(self.alpha) < 1.0:
            raise ValueError("use_cg=False needs |value| * alpha >= 1 for every stored entry "
                             "(smaller confidences can make a row's system indefinite)")
        # rows longer than this are summed chunk by chunk: fixed by the whole matrix, so that one GPU and any
        # number of ranks add the same partial sums in the same order
        chunk = E.auto_chunk(Cui.nnz)
        if self.devices is not None and len(self.devices) > 1:
            if callback is not None:
                raise ValueError("callback is not supported by a fit sharded over several devices")
            self._fit_sharded(Cui, chunk)
            self._dev_items = self._item_norms = self._serving = None
            return

        with self._lock, torch.cuda.device(dev):
            F = E.factor_width(self.factors)
            dCui = E.DeviceCSR.from_scipy(Cui, dev, scale=self.alpha)
            X = E.pad_factors(self.user_factors, F, dev)
            Y = E.pad_factors(self.item_factors, F, dev)
            if self.use_cg:
                eng = E.AlsEngine(dCui, None, X, Y, self.regularization, self.cg_steps, chunk_nnz=chunk)
            else:
                # every row with entries goes through the explicit-Gram kernel, which holds the Cholesky solve
                eng = E.AlsEngine(dCui, None, X, Y, self.regularization, self.SOLVE_EXACT, short_max=0, chunk_nnz=chunk)
            for it in range(self.iterations):
                eng.iteration()
                if callback is not None:
                    callback(it, None, None)
            self.user_factors = X[:, :self.factors].contiguous().cpu().numpy()
            self.item_factors = Y[:, :self.factors].contiguous().cpu().numpy()
            self._dev_items = None
            self._item_norms = None
            self._serving = None

    def _fit_sharded(self, Cui: sp.csr_matrix, chunk_nnz: int, timeout_s: float = 6 * 3600.0) -> None:
        """The fit on ``len(self.devices)`` ranks, one fresh child process per entry of ``devices`` (started
        with nothing of this process's GPU state: a process that has initialised the GPU must never be replaced
        by another program, and RCCL wants one process per device anyway).  Inputs and results travel through a
        scratch directory in shared memory; the children run ``fit_worker``.  Two entries that name the same
        card make the ranks talk over gloo through host memory (RCCL refuses two ranks on one device): the
        sharding is the same, so the factors are, bit for bit -- which is what the tests check on one GPU."""
        import json
        import shutil
        import socket
        import subprocess
        import sys
        import tempfile
        import time

        world = len(self.devices)
        backend = "nccl" if len(set(self.devices)) == world else "gloo"
        pkg_dir = os.path.dirname(os.path.abspath(__file__))
        scratch = tempfile.mkdtemp(prefix="mirec_fit_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        procs = []
        try:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            np.save(os.path.join(scratch, "indptr.npy"), np.ascontiguousarray(Cui.indptr, dtype=np.int64))
            np.save(os.path.join(scratch, "indices.npy"), np.ascontiguousarray(Cui.indices, dtype=np.int32))
            np.save(os.path.join(scratch, "data.npy"), np.ascontiguousarray(Cui.data, dtype=np.float32))
            np.save(os.path.join(scratch, "x0.npy"), np.ascontiguousarray(self.user_factors, dtype=np.float32))
            np.save(os.path.join(scratch, "y0.npy"), np.ascontiguousarray(self.item_factors, dtype=np.float32))
            with open(os.path.join(scratch, "job.json"), "w") as fh:
                json.dump({"devices": self.devices, "backend": backend, "port": port, "users": Cui.shape[0],
                           "items": Cui.shape[1], "factors": self.factors, "alpha": self.alpha,
                           "regularization": self.regularization, "cg_steps": self.cg_steps, "use_cg": bool(self.use_cg),
                           "iterations": self.iterations, "chunk_nnz": int(chunk_nnz)}, fh)
            env = dict(os.environ)
            env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(pkg_dir)] + [p for p in [env.get("PYTHONPATH")] if p])
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            module = __name__.rsplit(".", 1)[0] + ".fit_worker"
            for r in range(world):
                log = open(os.path.join(scratch, f"rank{r}.log"), "w")
                procs.append((subprocess.Popen([sys.executable, "-m", module, scratch, str(r)], env=env, stdout=log,
                                               stderr=subprocess.STDOUT), log))
            deadline = time.time() + timeout_s
            failed = None
            while failed is None and any(p.poll() is None for p, _ in procs):
                for r, (p, _) in enumerate(procs):
                    if p.poll() not in (None, 0):
                        failed = r
                if time.time() > deadline:
                    failed = -1
                time.sleep(0.05)
            if failed is None:
                failed = next((r for r, (p, _) in enumerate(procs) if p.returncode != 0), None)
            if failed is not None:
                for p, _ in procs:      # the ranks this call started, by handle
                    if p.poll() is None:
                        p.kill()
                for p, lg in procs:
                    p.wait()
                    lg.close()
                r = max(failed, 0)
                tail = open(os.path.join(scratch, f"rank{r}.log")).read()[-2000:]
                raise _lib.MirecError(("sharded fit timed out" if failed < 0 else f"rank {r} of the sharded fit failed")
                                      + f" (devices {self.devices}, backend {backend}):\n{tail}")
            for _, lg in procs:
                lg.close()
            self.user_factors = np.load(os.path.join(scratch, "x_out.npy"))
            self.item_factors = np.load(os.path.join(scratch, "y_out.npy"))
        finally:
            for p, _ in procs:
                if p.poll() is None:
                    p.kill()
                    p.wait()
            shutil.rmtree(scratch, ignore_errors=True)

    # -------------------------------------------------------------- inference
    def _device_items(self, dev: torch.device):
        if self._dev_items is None or self._dev_items.device != dev:
            W = E.scoring_width(self.factors, 1)
            self._dev_items = E.pad_factors(self.item_factors, W, dev)
        return self._dev_items

    def attach_user_items(self, user_items) -> None:
        """Keep the interaction matrix (one CSR row per user of the model) on the device, so that
        ``recommend(userid, None, ...)`` filters a user's liked items without receiving, slicing or uploading a
        row per request.  ``None`` detaches."""
        if user_items is None:
            self._attached = None
            return
        m = _check_csr(user_items)
        if self.user_factors is not None and m.shape[0] != self.user_factors.shape[0]:
            raise ValueError("attach_user_items needs one row per user of the model")
        if not m.has_sorted_indices:
            m = m.sorted_indices()
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            self._attached = (m, E.DeviceCSR.from_scipy(m, dev))

    def _serving_state(self, dev: torch.device) -> "_ServingState":
        """Everything a request needs, resident: both factor matrices at the scoring width, the item norms,
        and per-N output / workspace / pinned download buffers.  Call with the lock held."""
        st = self._serving
        if st is None or st.dev != dev:
            Yd = self._device_items(dev)
            st = _ServingState(dev, Yd, E.pad_factors(self.user_factors, Yd.shape[1], dev),
                               torch.from_numpy(self.item_norms).to(dev))
            self._serving = st
        return st

    def _one_request(self, q_ptr: int, N: int, mask_ptrs, norm_ptrs, dev: torch.device):
        """One query row that already sits on the device -> (ids, scores) numpy arrays of length N.  Two kernel
        launches and one 8N-byte download; nothing is uploaded or allocated once the buffers for this N exist."""
        lib = _lib.require()
        st = self._serving_state(dev)
        out, host, ws = st.buffers(int(N))
        ni, W = st.Y.shape
        mi, mx = mask_ptrs if mask_ptrs is not None else (0, 0)
        yn, qn = norm_ptrs if norm_ptrs is not None else (0, 0)
        base = out.data_ptr()
        _lib.check(lib.mirec_topk_small(q_ptr, 1, st.Y.data_ptr(), ni, W, mi, mx, yn, qn, int(N), base, base + 4 * N,
                                        ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "mirec_topk_small")
        host.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        arr = host.numpy()
        ids = arr[:N].copy()
        scores = arr[N:].view(np.float32).copy()
        return ids, scores

    def _recommend_one(self, uid: int, user_items, N: int, filter_liked: bool):
        """The serving call of the reference, ``model.recommend(user_idx, train_matrix[user_idx], N=N,
        filter_already_liked_items=True)`` (recommenders/als_recommender.py:181-186): one resident user row
        against all items."""
        if not 0 <= uid < self.user_factors.shape[0]:
            raise IndexError(f"userid {uid} is out of range")
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            st = self._serving_state(dev)
            keep = None
            mask_ptrs = None
            if filter_liked:
                if user_items is None:
                    if self._attached is None:
                        raise ValueError("user_items is needed to filter liked items (or attach_user_items first)")
                    d = self._attached[1]
                    mask_ptrs = (d.indptr.data_ptr() + 8 * uid, d.indices.data_ptr())
                else:
                    if not sp.isspmatrix_csr(user_items):
                        raise ValueError("user_items needs to be a CSR sparse matrix")
                    if user_items.shape[0] != 1:
                        raise ValueError("user_items must contain 1 row for every user in userids")
                    idx = np.sort(user_items.indices[user_items.indptr[0]:user_items.indptr[1]]).astype(np.int32)
                    keep = (torch.from_numpy(idx).to(dev) if idx.size else torch.zeros(1, dtype=torch.int32, device=dev),
                            torch.tensor([0, idx.size], dtype=torch.int64, device=dev))
                    mask_ptrs = (keep[1].data_ptr(), keep[0].data_ptr())
            ids, scores = self._one_request(st.X.data_ptr() + 4 * st.X.shape[1] * uid, N, mask_ptrs, None, dev)
            del keep
        if filter_liked:
            ids[scores == np.float32(_FILTERED)] = -1
        return ids, scores

    @property
    def item_norms(self) -> np.ndarray:
        if self._item_norms is None:
            n = np.linalg.norm(self.item_factors, axis=-1).astype(np.float32)
            n[n == 0] = 1e-10
            self._item_norms = n
        return self._item_norms

    def recommend(self, userid, user_items=None, N: int = 10, filter_already_liked_items: bool = True,
                  filter_items=None, recalculate_user: bool = False, items=None):
        """Top-N items for one user (scalar ``userid`` -> 1-D arrays) or a batch
        (array -> 2-D arrays); ``user_items`` holds one CSR row per user (None:
        the rows of the matrix given to ``attach_user_items``).  Liked
        items and ``filter_items`` are scored -FLT_MAX and, where fewer than N
        other items exist, fill the tail with id -1.  Returns (ids int32,
        scores float32), best first; ties by ascending item index."""
        if self.user_factors is None or self.item_factors is None:
            raise ValueError("model has not been fitted")
        if filter_items is not None and items is not None:
            raise ValueError("Can't specify both filter_items and items")
        scalar = np.isscalar(userid)
        if scalar and filter_items is None and items is None and not recalculate_user:
            return self._recommend_one(int(userid), user_items, int(N), bool(filter_already_liked_items))
        uids = np.atleast_1d(np.asarray(userid)).astype(np.int64)
        if uids.size and (uids.min() < 0 or uids.max() >= self.user_factors.shape[0]):
            raise IndexError("a userid is out of range")     # checked here: the device gathers without bounds checks
        if items is not None:
            items = np.asarray(items, dtype=np.int64).ravel()
            if items.size == 0 or items.min() < 0 or items.max() >= self.item_factors.shape[0]:
                raise IndexError("items must name at least one item, all of them in range")
        if recalculate_user:
            if user_items is None and self._attached is not None:
                user_items = self._attached[0][uids]
            if not sp.isspmatrix_csr(user_items) or user_items.shape[0] != len(uids):
                raise ValueError("recalculate_user needs user_items: a CSR matrix with 1 row for every user in userids")
        mask = None
        if filter_already_liked_items or filter_items is not None:
            if filter_already_liked_items:
                if user_items is None and self._attached is not None:
                    user_items = self._attached[0][uids]
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
            if items is not None:
                # the candidates are the given items, in the given order: the filter is told in their numbering
                mask = sp.csr_matrix(mask[:, items])
            if not mask.has_sorted_indices:
                mask = mask.sorted_indices()
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            st = self._serving_state(dev)
            Yd = st.Y
            if recalculate_user:
                Q = self._solve_rows(sp.csr_matrix(user_items), self.item_factors, dev, Yd.shape[1])
            else:
                Q = st.X.index_select(0, torch.from_numpy(uids).to(dev))
            if items is not None:
                Yd = Yd.index_select(0, torch.from_numpy(items).to(dev))
            dmask = None if mask is None else E.DeviceCSR.from_scipy(mask, dev)
            ids, scores = E.topk(Q, Yd, int(N), dmask)
            ids = _hide_filtered(ids, scores, dmask)
            ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        if items is not None:
            ids = np.where(ids >= 0, items[np.maximum(ids, 0)], -1).astype(np.int32)
        if scalar:
            return ids[0], scores[0]
        return ids, scores

    def _solve_rows(self, rows: sp.csr_matrix, other_host: np.ndarray, dev: torch.device, width: int) -> torch.Tensor:
        """Factors for the rows of ``rows`` from scratch, with the other side's factors fixed: per row the exact
        solution of (Y^T Y + reg I + sum_i (c_i - 1) y_i y_i^T) x = sum_i c_i y_i with the stored values as the
        confidences c_i -- implicit 0.7.2's ``recalculate_user`` / ``recalculate_item`` (its ``user_factor``,
        ``np.linalg.solve`` on ``user_linear_equation``), here one half-sweep of the Cholesky solver.  A row
        without entries gets zeros."""
        if rows.nnz and float(np.abs(rows.data).min()) < 1.0:
            raise ValueError("recalculating a factor needs |confidence| >= 1 for every stored entry "
                             "(smaller confidences can make the row's system indefinite)")
        rows = sp.csr_matrix(rows, dtype=np.float32)
        if not rows.has_sorted_indices:
            rows = rows.sorted_indices()
        if rows.shape[1] != other_host.shape[0]:
            raise ValueError("the rows have %d columns, the other side has %d rows of factors"
                             % (rows.shape[1], other_host.shape[0]))
        F = E.factor_width(self.factors)          # the width the solvers work at; `width`: what the caller scores at
        other = E.pad_factors(other_host, F, dev)
        d = E.DeviceCSR.from_scipy(rows, dev)
        G = E.gram(other, self.regularization)
        x = torch.zeros((rows.shape[0], F), dtype=torch.float32, device=dev)
        E.SweepPlan(d, F, short_max=0).sweep(x, other, G, self.SOLVE_EXACT)
        out = torch.zeros((rows.shape[0], width), dtype=torch.float32, device=dev)
        out[:, :self.factors] = x[:, :self.factors]
        return out

    def recalculate_user(self, userid, user_items) -> np.ndarray:
        """The factors of the users whose interaction rows are ``user_items`` (one row per entry of ``userid``),
        solved from scratch against the item factors -- for users the model was not fitted on, or whose history has
        changed.  ``userid`` only fixes the shape of the result (scalar -> one vector), as in implicit."""
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            x = self._solve_rows(sp.csr_matrix(user_items), self.item_factors, dev, self.factors).cpu().numpy()
        return x[0] if np.isscalar(userid) else x

    def recalculate_item(self, itemid, item_users) -> np.ndarray:
        """The same for items: ``item_users`` holds one row of user confidences per entry of ``itemid``."""
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            y = self._solve_rows(sp.csr_matrix(item_users), self.user_factors, dev, self.factors).cpu().numpy()
        return y[0] if np.isscalar(itemid) else y

    def recommend_all(self, user_items=None, N: int = 10, filter_already_liked_items: bool = True,
                      users: Optional[tuple] = None):
        """The batch form of ``recommend`` for users [u0, u1) (default: all):
        one fused pass, the score matrix never exists."""
        u0, u1 = users if users is not None else (0, self.user_factors.shape[0])
        dev = E.require_cuda(self.device)
        with self._lock, torch.cuda.device(dev):
            st = self._serving_state(dev)          # both factor matrices stay on the device between calls
            Yd = st.Y
            Q = st.X[u0:u1]
            dmask = None
            if filter_already_liked_items:
                att = self._attached
                if att is not None and (user_items is None or user_items is att[0]):
                    d = att[1]                     # the attached matrix: its rows as a view, nothing uploaded
                    if u1 - u0 > (1 << 17):
                        d.build_order()            # once per attached matrix: the sweep order of the all-users pass
                    dmask = d.rows(u0, u1)
                else:
                    if user_items is None:
                        raise ValueError("user_items is needed to filter liked items")
                    m = sp.csr_matrix(user_items)[u0:u1]
                    if not m.has_sorted_indices:
                        m = m.sorted_indices()
                    dmask = E.DeviceCSR.from_scipy(m, dev)
                    if u1 - u0 > (1 << 17):
                        dmask.build_order()
            n = Q.shape[0]
            if n <= (1 << 17):
                ids, scores = E.topk(Q, Yd, int(N), dmask)
                ids = _hide_filtered(ids, scores, dmask)
                return ids.cpu().numpy(), scores.cpu().numpy()
            # many users: chunk by chunk, each chunk's download overlapping the next chunk's kernels
            dl = E.ChunkedDownload(n, int(N), dev)
            order = E.norm_order(Yd) if dmask is None else None   # without a filter: one sweep order for all chunks
            for a, b in dl.ranges():
                # (the kernels index the filter's row pointers by query and read absolute offsets: a view will do)
                m = None if dmask is None else dmask.rows(a, b)
                ids, scores = E.topk(Q[a:b], Yd, int(N), m, path="sweep", item_order=order)
                dl.submit(_hide_filtered(ids, scores, m), scores, a)
            return dl.finish()

    def similar_items(self, itemid, N: int = 10, recalculate_item: bool = False, item_users=None,
                      filter_items=None, items=None):
        """Cosine-nearest items: scores = (Y . y) / |Y_i| / |y|, best first (the
        item itself comes first, as in implicit)."""
        if filter_items is not None and items is not None:
            raise ValueError("Can't specify both filter_items and items")
        scalar = np.isscalar(itemid)
        dev = E.require_cuda(self.device)
        if recalculate_item or filter_items is not None or items is not None:
            return self._similar_items_general(itemid, int(N), recalculate_item, item_users, filter_items, items, dev)
        if scalar:
            iid = int(itemid)
            if not 0 <= iid < self.item_factors.shape[0]:
                raise IndexError(f"itemid {iid} is out of range")
            with self._lock, torch.cuda.device(dev):
                st = self._serving_state(dev)
                return self._one_request(st.Y.data_ptr() + 4 * st.Y.shape[1] * iid, int(N), None,
                                         (st.norms.data_ptr(), st.norms.data_ptr() + 4 * iid), dev)
        iids = np.atleast_1d(np.asarray(itemid)).astype(np.int64)
        if iids.size and (iids.min() < 0 or iids.max() >= self.item_factors.shape[0]):
            raise IndexError("an itemid is out of range")
        norms = self.item_norms
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

    def _similar_items_general(self, itemid, N, recalculate_item, item_users, filter_items, items, dev):
        """``similar_items`` with implicit's optional arguments: the query vector solved again from ``item_users``
        (its norm with it), ``filter_items`` left out of the answer, or the candidates restricted to ``items``
        (ids come back in the catalogue's numbering)."""
        scalar = np.isscalar(itemid)
        iids = np.atleast_1d(np.asarray(itemid)).astype(np.int64)
        if iids.size and (iids.min() < 0 or iids.max() >= self.item_factors.shape[0]):
            raise IndexError("an itemid is out of range")
        norms = self.item_norms
        if items is not None:
            items = np.asarray(items, dtype=np.int64).ravel()
            if items.size == 0 or items.min() < 0 or items.max() >= self.item_factors.shape[0]:
                raise IndexError("items must name at least one item, all of them in range")
        if recalculate_item:
            if not sp.isspmatrix_csr(item_users) or item_users.shape[0] != len(iids):
                raise ValueError("recalculate_item needs item_users: a CSR matrix with 1 row for every item in itemid")
        with self._lock, torch.cuda.device(dev):
            st = self._serving_state(dev)
            Yd = st.Y
            if recalculate_item:
                Q = self._solve_rows(sp.csr_matrix(item_users), self.user_factors, dev, Yd.shape[1])
                qn_host = np.linalg.norm(Q[:, :self.factors].cpu().numpy(), axis=-1).astype(np.float32)
                qn_host[qn_host == 0] = 1e-10
            else:
                Q = Yd.index_select(0, torch.from_numpy(iids).to(dev))
                qn_host = np.ascontiguousarray(norms[iids])
            yn = torch.from_numpy(norms if items is None else np.ascontiguousarray(norms[items])).to(dev)
            if items is not None:
                Yd = Yd.index_select(0, torch.from_numpy(items).to(dev))
            dmask = None
            if filter_items is not None:
                fi = np.unique(np.asarray(filter_items, dtype=np.int64))
                fi = fi[(fi >= 0) & (fi < Yd.shape[0])]
                m = sp.csr_matrix((np.ones(len(fi) * len(iids), dtype=np.float32),
                                   (np.repeat(np.arange(len(iids)), len(fi)), np.tile(fi, len(iids)))),
                                  shape=(len(iids), Yd.shape[0]))
                m.sort_indices()
                dmask = E.DeviceCSR.from_scipy(m, dev)
            ids, scores = E.topk(Q.contiguous(), Yd.contiguous(), N, dmask, y_norm=yn,
                                 q_norm=torch.from_numpy(qn_host).to(dev))
            ids = _hide_filtered(ids, scores, dmask)
            ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        if items is not None:
            ids = np.where(ids >= 0, items[np.maximum(ids, 0)], -1).astype(np.int32)
        if scalar:
            return ids[0], scores[0]
        return ids, scores
