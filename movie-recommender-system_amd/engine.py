"""Device-side plumbing between PyTorch-ROCm tensors and the C ABI.

PyTorch is used for what it is good at here -- device memory, streams and (in
``dist.py``) RCCL -- while every arithmetic step of the hot path is a libmirec
call.  Classes:

``DeviceCSR``   a CSR matrix resident in HBM (int64 indptr, int32 indices, fp32 data)
``SweepPlan``   the chunk plan + partial-Gram workspace of one half-sweep
``AlsEngine``   factor matrices, plans and workspaces of one fit; one
                ``iteration()`` = user half-sweep + item half-sweep, as in
                implicit's ``fit`` loop (reference call sites ``als/als.py:174``,
                ``recommenders/als_recommender.py:136``)
"""

from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


def _stream_ptr() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def require_cuda(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.MirecError("no HIP device is visible: the recommender core runs on MI355X only (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise _lib.MirecError(f"device {dev} is not a HIP device")
    return dev


def factor_width(f: int, for_scoring: bool = False) -> int:
    w = _lib.require().mirec_factor_width(int(f), 1 if for_scoring else 0)
    if w < 0:
        raise _lib.MirecError(f"factor width {f} exceeds what the kernels are built for (192)")
    return int(w)


def scoring_width(f: int, k: int) -> int:
    """Padded width for a top-k pass: the smallest supported width, moved up to a
    multiple of 16 when that lets k <= 24 run on the bf16-plane kernel (zero
    columns do not change a score: fma(0, 0, s) == s)."""
    w = factor_width(f, for_scoring=True)
    if k <= 24 and w <= 128 and w % 16:
        w = factor_width(w + 8, for_scoring=True)
    return w


def pad_factors(a: np.ndarray, width: int, device: torch.device) -> torch.Tensor:
    """Upload [n, f] fp32 as a zero-padded [n, width] device matrix."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    n, f = a.shape
    out = torch.zeros((n, width), dtype=torch.float32, device=device)
    if n:
        out[:, :f] = torch.from_numpy(a).to(device)
    return out


class DeviceCSR:
    """CSR matrix in HBM.  ``indptr`` holds offsets into this object's own
    ``indices``/``data`` (it starts at 0)."""

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, data: torch.Tensor, shape: Tuple[int, int]):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32 and data.dtype == torch.float32
        self.indptr, self.indices, self.data = indptr, indices, data
        self.shape = (int(shape[0]), int(shape[1]))
        self.nnz = int(indices.numel())
        self.order = None   # (item_order, blocks, block_len) once build_order has run: see there

    def build_order(self):
        """Prepare this matrix as the liked-item filter of all-users top-K passes: the items in descending number
        of interactions, and the filter as one list per 32 rows (the queries of a wavefront) in positions of that
        order.  ``topk`` then sweeps the items in that order -- same results, a fraction of the work.  Once per
        matrix (it does not depend on the factors); row views made by ``rows`` share it.  Returns False, and
        changes nothing, for catalogues too large for the list kernel."""
        lib = _lib.require()
        nu, ni = self.shape
        ws_bytes = int(lib.mirec_topk_order_workspace_bytes(ni))
        if self.order is not None or ws_bytes == 0:
            return self.order is not None
        dev = self.indices.device
        item_order = torch.empty(ni, dtype=torch.int32, device=dev)
        blocks = torch.empty((max(self.nnz, 1), 2), dtype=torch.int32, device=dev)
        block_len = torch.empty(max((nu + 31) // 32, 1), dtype=torch.int32, device=dev)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.mirec_topk_order_build(_ptr(self.indptr), _ptr(self.indices), nu, self.nnz, ni,
                                              _ptr(item_order), _ptr(blocks), _ptr(block_len), _ptr(ws), ws_bytes,
                                              _stream_ptr()), "mirec_topk_order_build")
        self.order = (item_order, blocks, block_len)
        return True

    def rows(self, a: int, b: int) -> "DeviceCSR":
        """Rows [a, b) as a view: the kernels index the row pointers by query and read absolute offsets, so
        nothing is copied (not a matrix of its own: ``indptr`` does not start at 0).  The view keeps the sweep
        order when it starts at a block of 32 rows."""
        v = DeviceCSR(self.indptr[a:b + 1], self.indices, self.data, (b - a, self.shape[1]))
        if self.order is not None and a % 32 == 0:
            v.order = (self.order[0], self.order[1], self.order[2][a // 32:])
        return v

    @classmethod
    def from_scipy(cls, m: sp.csr_matrix, device: torch.device, scale: float = 1.0) -> "DeviceCSR":
        m = sp.csr_matrix(m)
        indptr = torch.from_numpy(np.ascontiguousarray(m.indptr, dtype=np.int64)).to(device)
        indices = torch.from_numpy(np.ascontiguousarray(m.indices, dtype=np.int32)).to(device)
        data = np.ascontiguousarray(m.data, dtype=np.float32)
        if scale != 1.0:
            data = data * np.float32(scale)
        return cls(indptr, indices, torch.from_numpy(data).to(device), m.shape)

    @classmethod
    def from_events(cls, user_idx: torch.Tensor, item_idx: torch.Tensor, shape: Tuple[int, int],
                    with_item_counts: bool = False):
        """Binary CSR of the unique (user, item) pairs of an event list (int32
        device tensors): duplicates collapse, every stored value is 1.0, rows
        have ascending columns.  With ``with_item_counts`` also returns the
        number of users per item (int32 tensor)."""
        lib = _lib.require()
        assert user_idx.dtype == torch.int32 and item_idx.dtype == torch.int32
        assert user_idx.is_contiguous() and item_idx.is_contiguous() and user_idx.numel() == item_idx.numel()
        dev = user_idx.device
        n, (nu, ni) = int(user_idx.numel()), shape
        indptr = torch.empty(nu + 1, dtype=torch.int64, device=dev)
        indices = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        data = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
        counts = torch.empty(max(ni, 1), dtype=torch.int32, device=dev) if with_item_counts else None
        nnz = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.empty(max(int(lib.mirec_events_to_csr_workspace_bytes(n, nu)), 1), dtype=torch.uint8, device=dev)
        _lib.check(lib.mirec_events_to_csr(_ptr(user_idx), _ptr(item_idx), n, nu, ni, _ptr(indptr), _ptr(indices),
                                           _ptr(data), _ptr(counts), _ptr(nnz), _ptr(ws), ws.numel(), _stream_ptr()),
                   "mirec_events_to_csr")
        k = int(nnz.item())
        csr = cls(indptr, indices[:k].clone(), data[:k].clone(), (nu, ni))
        return (csr, counts[:ni]) if with_item_counts else csr

    def transpose(self) -> "DeviceCSR":
        """``.T.tocsr()`` on the device (stable sort by column)."""
        lib = _lib.require()
        rows, cols = self.shape
        dev = self.indptr.device
        ws_bytes = lib.mirec_csr_transpose_workspace_bytes(rows, cols, self.nnz)
        ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=dev)
        t_indptr = torch.empty(cols + 1, dtype=torch.int64, device=dev)
        t_indices = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        t_data = torch.empty(self.nnz, dtype=torch.float32, device=dev)
        _lib.check(lib.mirec_csr_transpose(_ptr(self.indptr), _ptr(self.indices), _ptr(self.data), rows, cols, self.nnz,
                                           _ptr(t_indptr), _ptr(t_indices), _ptr(t_data), _ptr(ws), ws_bytes,
                                           _stream_ptr()), "mirec_csr_transpose")
        # `ws` may be freed by the caching allocator only after the kernels that use it: same stream, so safe.
        return DeviceCSR(t_indptr, t_indices, t_data, (cols, rows))

    def row_slice(self, r0: int, r1: int) -> "DeviceCSR":
        """Rows [r0, r1) as a CSR of their own (copies the pointer array only)."""
        s = int(self.indptr[r0].item())
        e = int(self.indptr[r1].item())
        return DeviceCSR((self.indptr[r0:r1 + 1] - s).contiguous(), self.indices[s:e], self.data[s:e],
                         (r1 - r0, self.shape[1]))

    def to_scipy(self) -> sp.csr_matrix:
        return sp.csr_matrix((self.data.cpu().numpy(), self.indices.cpu().numpy(), self.indptr.cpu().numpy()),
                             shape=self.shape)


def gram(Y: torch.Tensor, reg: float, out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
         absmax: Optional[torch.Tensor] = None):
    """G = Y^T Y + reg I for a padded device factor matrix; ``absmax`` (one device float) receives the largest
    |entry| of Y, found on the same pass."""
    lib = _lib.require()
    n, f = Y.shape
    need = lib.mirec_gram_workspace_bytes(n, f)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=Y.device)
    if out is None:
        out = torch.empty((f, f), dtype=torch.float32, device=Y.device)
    _lib.check(lib.mirec_gram(_ptr(Y), n, f, float(reg), _ptr(out), _ptr(absmax), _ptr(ws), ws.numel(), _stream_ptr()),
               "mirec_gram")
    return out


def auto_chunk(nnz: int) -> int:
    """The chunk length the library chooses for a matrix of ``nnz`` stored entries (rows longer than it are cut
    into chunks whose partial sums are added in chunk order).  A sharded fit passes the value of the WHOLE
    matrix to every rank, so that the sums, and with them the result bits, do not depend on the world size."""
    return int(min(max((int(nnz) // 16384 + 31) // 32 * 32, 4096), 1 << 20))


# The matrix-free kernel re-reads a row's gathered factor rows on every CG step.  That is cheap while the
# gathered table sits in L2 / Infinity Cache and a 4x HBM bill once it does not, so rows of any length go
# to it only below this table size; above it only rows that fit its LDS cache do.
SHORT_TABLE_BYTES = 96 << 20
ALL_ROWS = (1 << 31) - 1


class SweepPlan:
    """Chunk plan of one CSR row range for a given factor width.

    ``short_max``: rows with at most this many stored entries are solved by the
    matrix-free wave-per-row kernel, the others by the explicit-Gram kernel
    (None = choose from the size of the gathered table)."""

    def __init__(self, csr: DeviceCSR, f: int, chunk_nnz: int = 0, short_max: Optional[int] = None,
                 conf_max: float = 0.0):
        """``conf_max``: the largest |confidence| of the whole matrix (0 = let every sweep find its own
        rows' with one pass over their values); it fixes the power of two that scales the explicit-Gram
        kernel's fp16 operands, so ranks of a sharded fit should all pass the matrix-wide value."""
        lib = _lib.require()
        self.csr, self.f = csr, int(f)
        self.conf_max = float(conf_max)
        nrows = csr.shape[0]
        dev = csr.indptr.device
        if short_max is None:
            table_bytes = csr.shape[1] * self.f * 4
            # Measured at f = 128 (tools/probes/short_max_sweep.sh): with the gathered table on chip (L2 /
            # Infinity Cache; 1 M users x 27 k items) the user half-sweep takes 14.6 / 9.7 / 7.3 / 6.4 / 6.3 / 6.4 ms
            # for limits of 32 / 64 / 128 / 256 / 512 / 1024 entries; with a larger table (2 M users x 200 k items,
            # 102 MB, mean 187 entries) the four passes of gathers cost more and the explicit Gram takes over
            # earlier: 60.7 / 52.3 / 48.4 / 51.7 / 57.0 ms for 64 / 128 / 256 / 512 / 1024.
            # The Gram's cost per entry grows with f^2, a gather's with f: the limits scale with f.
            limit = 512 if table_bytes <= SHORT_TABLE_BYTES else 256
            short_max = max(limit * self.f // 128, int(lib.mirec_als_short_cache_rows(self.f)), 32)
        self.short_max = int(short_max)
        nbytes = lib.mirec_als_plan_bytes(nrows, csr.nnz, int(chunk_nnz))
        self.plan = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        chunk = ctypes.c_int32(int(chunk_nnz))
        self.counts = (ctypes.c_int64 * 4)()
        _lib.check(lib.mirec_als_plan_build(_ptr(csr.indptr), 0, nrows, csr.nnz, self.f, _ptr(self.plan),
                                            self.plan.numel(), ctypes.byref(chunk), self.short_max, self.counts,
                                            _stream_ptr()), "mirec_als_plan_build")
        self.chunk_nnz = int(chunk.value)
        ws_bytes = lib.mirec_als_sweep_workspace_bytes(self.counts, self.f)
        self.ws = torch.empty(max(int(ws_bytes), 1), dtype=torch.uint8, device=dev)

    @property
    def n_chunks(self) -> int:
        return int(self.counts[0])

    @property
    def n_split_rows(self) -> int:
        return int(self.counts[1])

    @property
    def n_short_rows(self) -> int:
        return int(self.counts[3])

    def sweep(self, X_rows: torch.Tensor, Y: torch.Tensor, G: torch.Tensor, cg_steps: int = 3,
              phase: str = "both", y_rows: Optional[int] = None, y_absmax: Optional[torch.Tensor] = None) -> None:
        """Solve the plan's rows in place: ``X_rows`` is the [nrows, f] block of
        the factor matrix that belongs to these CSR rows.  ``phase`` lets a
        profiler bracket the kernels of a sweep separately: "short" launches
        only the matrix-free kernel, "main" only the Gram+CG kernel, "finalize"
        only the split-row reduction (which must follow "main" on the stream).
        ``y_rows`` overrides the row count of ``Y`` passed to the library (0 =
        "not known", which selects 64-bit addressing of the gathered table).
        ``y_absmax``: the largest |entry| of ``Y`` as ``gram`` returns it (one
        device float; None: the library bounds it from the diagonal of G)."""
        lib = _lib.require()
        csr = self.csr
        assert X_rows.shape == (csr.shape[0], self.f) and X_rows.is_contiguous()
        assert Y.shape[1] == self.f and Y.is_contiguous() and Y.shape[0] >= csr.shape[1]
        counts, short_max = self.counts, self.short_max
        if phase != "both":
            counts = (ctypes.c_int64 * 4)(self.counts[0] if phase == "main" else 0,
                                          self.counts[1] if phase == "finalize" else 0, self.counts[2],
                                          self.counts[3] if phase == "short" else 0)
            if phase != "short":
                short_max = -1
        _lib.check(lib.mirec_als_half_sweep(_ptr(csr.indptr), _ptr(csr.indices), _ptr(csr.data), 0, csr.shape[0],
                                            csr.nnz, _ptr(X_rows), _ptr(Y), int(Y.shape[0] if y_rows is None else y_rows), _ptr(G),
                                            _ptr(y_absmax), self.f, int(cg_steps),
                                            _ptr(self.plan), self.chunk_nnz, short_max, counts, self.conf_max,
                                            _ptr(self.ws), self.ws.numel(), _stream_ptr()), "mirec_als_half_sweep")


class AlsEngine:
    """All device state of one ALS fit on one GPU.

    ``user_blocks`` (row ranges, solved in order) / ``item_rows`` restrict the
    rows this engine solves (the shard of one rank); the factor matrices are
    always held whole, the interaction matrix only as far as those rows go.
    """

    def __init__(self, Cui: Optional[DeviceCSR], Ciu: Optional[DeviceCSR], X: torch.Tensor, Y: torch.Tensor,
                 regularization: float, cg_steps: int = 3,
                 user_blocks: Optional[Sequence[Tuple[int, int]]] = None,
                 item_rows: Optional[Tuple[int, int]] = None, chunk_nnz: int = 0, short_max: Optional[int] = None,
                 user_shards: Optional[Sequence[DeviceCSR]] = None, item_shard: Optional[DeviceCSR] = None,
                 conf_max: Optional[float] = None):
        """``Cui`` / ``Ciu``: the whole matrix and its transpose (``Ciu`` None: transposed here), sliced to
        ``user_blocks`` / ``item_rows``.  A rank that holds only its share passes instead ``user_shards`` (one
        CSR per entry of ``user_blocks``, rows [a, b) of the matrix) and ``item_shard`` (rows ``item_rows`` of
        the transpose) and leaves ``Cui`` / ``Ciu`` None."""
        self.f = int(X.shape[1])
        assert Y.shape[1] == self.f
        self.X, self.Y = X, Y
        self.reg, self.cg_steps = float(regularization), int(cg_steps)
        users, items = int(X.shape[0]), int(Y.shape[0])
        self.user_blocks = [tuple(b) for b in (user_blocks if user_blocks is not None else [(0, users)])]
        self.item_rows = tuple(item_rows) if item_rows is not None else (0, items)
        if user_shards is None:
            assert Cui is not None and Cui.shape == (users, items)
            user_shards = [Cui if (a, b) == (0, users) else Cui.row_slice(a, b) for a, b in self.user_blocks]
        if item_shard is None:
            if Ciu is None:
                Ciu = Cui.transpose()
            item_shard = Ciu if self.item_rows == (0, items) else Ciu.row_slice(*self.item_rows)
        assert len(user_shards) == len(self.user_blocks)
        if conf_max is None:
            # the largest |confidence| of what this engine holds (the whole matrix, unless a sharded driver says otherwise)
            conf_max = max([float(m.data.abs().max()) for m in list(user_shards) + [item_shard] if m.nnz] + [0.0])
        self.conf_max = float(conf_max)
        self.plans_u = []
        for (a, b), cu in zip(self.user_blocks, user_shards):
            assert cu.shape == (b - a, items), "a user shard must hold exactly the rows of its block"
            self.plans_u.append(SweepPlan(cu, self.f, chunk_nnz, short_max, self.conf_max) if b > a else None)
        assert item_shard.shape == (self.item_rows[1] - self.item_rows[0], users)
        self.Ciu_local = item_shard
        self.plan_i = SweepPlan(item_shard, self.f, chunk_nnz, short_max, self.conf_max)
        lib = _lib.require()
        need = max(lib.mirec_gram_workspace_bytes(users, self.f), lib.mirec_gram_workspace_bytes(items, self.f))
        self.gram_ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=X.device)
        self.G = torch.empty((self.f, self.f), dtype=torch.float32, device=X.device)
        self.absmax = torch.zeros(1, dtype=torch.float32, device=X.device)   # largest |entry| of the table G is the Gram of

    # Every launch goes through _run so that a benchmark can bracket each kernel
    # with events on the launching stream (self.events = [] switches it on).
    events = None

    def _run(self, name: str, fn) -> None:
        if self.events is None:
            fn()
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.events.append((name, a, b))

    def _sweep(self, tag: str, plan: "SweepPlan", rows: torch.Tensor, other: torch.Tensor) -> None:
        if self.events is None:
            plan.sweep(rows, other, self.G, self.cg_steps, y_absmax=self.absmax)
            return
        if plan.n_short_rows:
            self._run(tag + "_short", lambda: plan.sweep(rows, other, self.G, self.cg_steps, "short", y_absmax=self.absmax))
        if plan.n_chunks:
            self._run(tag + "_main", lambda: plan.sweep(rows, other, self.G, self.cg_steps, "main", y_absmax=self.absmax))
        if plan.n_split_rows:
            self._run(tag + "_finalize", lambda: plan.sweep(rows, other, self.G, self.cg_steps, "finalize", y_absmax=self.absmax))

    @property
    def plan_u(self) -> "SweepPlan":
        """The user-side plan of an engine with a single user block."""
        assert len(self.plans_u) == 1
        return self.plans_u[0]

    def resident_matrix_bytes(self) -> int:
        """Bytes of interaction-matrix storage (row pointers, indices, values; both orientations) this engine
        keeps on its device."""
        mats = [p.csr for p in self.plans_u if p is not None] + [self.Ciu_local]
        return sum(m.indptr.numel() * 8 + m.indices.numel() * 4 + m.data.numel() * 4 for m in mats)

    def user_totals(self) -> Tuple[int, int, int, int]:
        """(rows, stored entries, rows of the matrix-free kernel, chunks) over this engine's user blocks."""
        ps = [p for p in self.plans_u if p is not None]
        return (sum(p.csr.shape[0] for p in ps), sum(p.csr.nnz for p in ps), sum(p.n_short_rows for p in ps),
                sum(p.n_chunks for p in ps))

    def user_sweep_block(self, j: int) -> None:
        """Solve user block j; block 0 first refreshes the Gram of Y."""
        if j == 0:
            self._run("gram_items", lambda: gram(self.Y, self.reg, self.G, self.gram_ws, self.absmax))
        plan = self.plans_u[j]
        if plan is not None:
            u0, u1 = self.user_blocks[j]
            self._sweep("user", plan, self.X[u0:u1], self.Y)

    def user_sweep(self) -> None:
        for j in range(len(self.user_blocks)):
            self.user_sweep_block(j)

    def item_sweep(self) -> None:
        self._run("gram_users", lambda: gram(self.X, self.reg, self.G, self.gram_ws, self.absmax))
        i0, i1 = self.item_rows
        self._sweep("item", self.plan_i, self.Y[i0:i1], self.X)

    def iteration(self) -> None:
        self.user_sweep()
        self.item_sweep()

    def event_times_ms(self) -> dict:
        """Sum and count of the recorded intervals per kernel name (call after a
        synchronize)."""
        out = {}
        for name, a, b in self.events or []:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + a.elapsed_time(b), n + 1)
        return out


# Up to this many queries the items are split over the workgroups (mirec_topk_small); beyond it the queries
# are (mirec_topk).  Measured crossover at 27 k items, f = 128, k = 20 (tools/probes/topk_small_crossover.py).
ORDER_MIN_QUERIES = 32768     # from this many queries on, a filter without a sweep order gets one made
SMALL_TOPK_QUERIES = 2048
SMALL_TOPK_WS_LIMIT = 512 << 20   # ... unless the tiles' partial lists would need more than this


def topk_small(Q: torch.Tensor, Y: torch.Tensor, k: int, mask: Optional[DeviceCSR] = None,
               y_norm: Optional[torch.Tensor] = None, q_norm: Optional[torch.Tensor] = None,
               n_items: Optional[int] = None, ws: Optional[torch.Tensor] = None,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """``topk`` for a handful of queries (the serving form: one user per request): the items are split over
    the workgroups.  Same results as ``topk``.  ``ws`` / ``out`` let a caller that serves many requests reuse
    its buffers."""
    lib = _lib.require()
    nq, f = Q.shape
    ni = int(Y.shape[0] if n_items is None else n_items)
    assert Y.shape[1] == f and Q.is_contiguous() and Y.is_contiguous()
    if mask is not None:
        assert mask.shape[0] == nq, "one mask row per query"
    if out is None:
        out = (torch.empty((nq, k), dtype=torch.int32, device=Q.device),
               torch.empty((nq, k), dtype=torch.float32, device=Q.device))
    need = int(lib.mirec_topk_small_workspace_bytes(nq, ni, int(k)))
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=Q.device)
    _lib.check(lib.mirec_topk_small(_ptr(Q), nq, _ptr(Y), ni, f, _ptr(None if mask is None else mask.indptr),
                                    _ptr(None if mask is None else mask.indices), _ptr(y_norm), _ptr(q_norm), int(k),
                                    _ptr(out[0]), _ptr(out[1]), _ptr(ws), ws.numel(), _stream_ptr()),
               "mirec_topk_small")
    return out


def topk(Q: torch.Tensor, Y: torch.Tensor, k: int, mask: Optional[DeviceCSR] = None,
         y_norm: Optional[torch.Tensor] = None, q_norm: Optional[torch.Tensor] = None,
         n_items: Optional[int] = None, workspace: bool = True, path: str = "auto",
         item_order: Optional[torch.Tensor] = None):
    """Fused score + top-k of every row of Q against the first n_items rows of Y.
    Returns device tensors (ids int32 [nq, k], scores fp32 [nq, k]).  With
    ``workspace`` the library's scratch (for the bf16-plane kernel, where it
    applies) is taken from torch's caching allocator; results are identical
    either way.  ``path``: "sweep" = the all-users kernels (queries split over
    the workgroups), "small" = the per-request kernels (items split), "auto" =
    by the number of queries.  ``item_order`` (without a filter): a sweep
    order for the all-users kernel (``norm_order``), for a caller that scores
    chunk after chunk against the same items; without it the library makes
    one inside every call."""
    lib = _lib.require()
    nq, f = Q.shape
    ni = int(Y.shape[0] if n_items is None else n_items)
    assert Y.shape[1] == f and Q.is_contiguous() and Y.is_contiguous()
    assert path in ("auto", "sweep", "small")
    if path == "auto":
        small = 0 < nq <= SMALL_TOPK_QUERIES and k <= 224 and \
            int(lib.mirec_topk_small_workspace_bytes(nq, ni, int(k))) <= SMALL_TOPK_WS_LIMIT
        path = "small" if small else "sweep"
    if path == "small":
        return topk_small(Q, Y, k, mask, y_norm, q_norm, n_items)
    if mask is not None:
        assert mask.shape[0] == nq, "one mask row per query"
    ids = torch.empty((nq, k), dtype=torch.int32, device=Q.device)
    scores = torch.empty((nq, k), dtype=torch.float32, device=Q.device)
    ws = None
    ws_bytes = 0
    if workspace and y_norm is None and q_norm is None:
        ws_bytes = int(lib.mirec_topk_workspace_bytes(nq, ni, f, int(k)))
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Q.device)
    if ws is not None and mask is not None and mask.order is None and nq >= ORDER_MIN_QUERIES and mask.shape[1] == ni:
        mask.build_order()    # worth its cost within this one pass; a matrix that is kept keeps the order too
    if ws is not None and mask is not None and mask.order is not None and mask.shape[1] == ni:
        item_order, blocks, block_len = mask.order
        _lib.check(lib.mirec_topk_ordered(_ptr(Q), nq, _ptr(Y), ni, f, _ptr(mask.indptr), _ptr(mask.indices),
                                          _ptr(item_order), _ptr(blocks), _ptr(block_len), int(k), _ptr(ids),
                                          _ptr(scores), _ptr(ws), ws_bytes, _stream_ptr()), "mirec_topk_ordered")
        return ids, scores
    if ws is not None and mask is None and item_order is not None:
        assert item_order.dtype == torch.int32 and int(item_order.numel()) == ni
        _lib.check(lib.mirec_topk_ordered(_ptr(Q), nq, _ptr(Y), ni, f, None, None, _ptr(item_order), None, None, int(k),
                                          _ptr(ids), _ptr(scores), _ptr(ws), ws_bytes, _stream_ptr()), "mirec_topk_ordered")
        return ids, scores
    _lib.check(lib.mirec_topk(_ptr(Q), nq, _ptr(Y), ni, f, _ptr(None if mask is None else mask.indptr),
                              _ptr(None if mask is None else mask.indices), _ptr(y_norm), _ptr(q_norm), int(k),
                              _ptr(ids), _ptr(scores), _ptr(ws), ws_bytes, _stream_ptr()), "mirec_topk")
    return ids, scores


def norm_order(Y: torch.Tensor, n_items: Optional[int] = None) -> torch.Tensor:
    """The items by descending row norm (int32, position -> item): a sweep order for ``topk`` without a filter."""
    ni = int(Y.shape[0] if n_items is None else n_items)
    return torch.argsort(Y[:ni].norm(dim=1), descending=True, stable=True).to(torch.int32)


class ChunkedDownload:
    """Results of a scoring pass over many queries, brought to host memory while the pass is still running.

    A pass is cut into chunks of queries; each chunk's (ids, scores) go from the device into one of two pinned
    staging buffers on a copy stream of their own and from there into the caller's numpy arrays, so the download
    of one chunk and the host-side copy of the one before overlap the kernels of the next.  (One 80 MB
    ``tensor.cpu()`` into pageable memory takes 18 ms on an MI355X host, as long as the whole scoring kernel of
    a million users; through pinned staging it is 1.6 ms plus a 6 ms host copy that hides behind the GPU.)"""

    _pinned = {}   # (rows, k) -> two (ids, scores) pinned pairs, reused from call to call

    def __init__(self, n: int, k: int, device: torch.device, chunk: int = 1 << 17):
        self.n, self.k, self.chunk = int(n), int(k), int(min(chunk, max(n, 1)))
        self.ids = np.empty((self.n, self.k), dtype=np.int32)
        self.scores = np.empty((self.n, self.k), dtype=np.float32)
        key = (self.chunk, self.k)
        if key not in self._pinned:
            self._pinned[key] = [(torch.empty((self.chunk, self.k), dtype=torch.int32).pin_memory(),
                                  torch.empty((self.chunk, self.k), dtype=torch.float32).pin_memory()) for _ in range(2)]
        self.staging = self._pinned[key]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pending = [None, None]   # per staging slot: (event, row0, rows, device tensors kept alive)
        self.slot = 0

    def ranges(self):
        return [(a, min(a + self.chunk, self.n)) for a in range(0, self.n, self.chunk)]

    def _drain(self, slot: int) -> None:
        p = self.pending[slot]
        if p is None:
            return
        ev, row0, rows, _keep = p
        ev.synchronize()
        hi, hs = self.staging[slot]
        self.ids[row0:row0 + rows] = hi[:rows].numpy()
        self.scores[row0:row0 + rows] = hs[:rows].numpy()
        self.pending[slot] = None

    def submit(self, dev_ids: torch.Tensor, dev_scores: torch.Tensor, row0: int) -> None:
        """Queue the download of one chunk's results (device tensors [rows, k] just produced on the current stream)."""
        rows = int(dev_ids.shape[0])
        slot = self.slot
        self._drain(slot)                       # the staging buffer is free again
        done = torch.cuda.Event()
        done.record()                           # the kernels of this chunk, on the compute stream
        hi, hs = self.staging[slot]
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(done)
            hi[:rows].copy_(dev_ids, non_blocking=True)
            hs[:rows].copy_(dev_scores, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self.pending[slot] = (ev, int(row0), rows, (dev_ids, dev_scores))
        self.slot ^= 1

    def finish(self):
        self._drain(self.slot)
        self._drain(self.slot ^ 1)
        return self.ids, self.scores
