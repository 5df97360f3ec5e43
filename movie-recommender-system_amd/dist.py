"""Row-sharded ALS over the GPUs of one node (one process per GPU).

Rows of a half-sweep are independent given the other side's factors and its
Gram, so both half-sweeps shard by rows; every rank keeps both factor matrices
whole and the freshly solved rows are exchanged with all-gathers (RCCL over
xGMI when the process group's backend is "nccl").  The Gram of the gathered
matrix is recomputed locally by every rank instead of all-reduced, so every
rank -- and every world size -- works from identical bits.  Scoring needs no
exchange: each rank scores its users.

The two sides are exchanged differently because their volumes differ by the
ratio of users to items:

* users (the big matrix) are owned block-cyclically: the row range is cut
  into ``blocks`` slabs of ``world`` equal blocks, rank r owns block r of every
  slab.  A slab is gathered in place (each rank's block already sits at its
  final position in the padded matrix, so there is no staging copy) as soon as
  its block is solved, while the next block is being solved: all but the last
  slab's exchange hides behind compute.
* items (small, Zipf-skewed) are owned as contiguous ranges balanced by stored
  entries and exchanged with one padded all-gather.

Storage is sharded like the solves: a rank holds the interaction rows of its
user blocks and -- after one all-to-all at set-up (``shard_item_rows``) -- the
rows of the transposed matrix for its item range, i.e. about 1/world of the
stored entries in either orientation; only the factor matrices are whole.

The driver takes the two local solves as callables, so the same partition and
exchange logic runs on CPU tensors over gloo in tests.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def partition_rows(indptr, world: int) -> List[int]:
    """Boundaries b[0..world] of contiguous row blocks with near-equal stored
    entries: block r = rows [b[r], b[r+1])."""
    if isinstance(indptr, torch.Tensor):
        indptr = indptr.detach().cpu().numpy()
    indptr = np.asarray(indptr, dtype=np.int64)
    nrows = len(indptr) - 1
    base = indptr[0]
    nnz = int(indptr[-1] - base)
    bounds = [0]
    for r in range(1, world):
        target = base + (nnz * r) // world
        b = int(np.searchsorted(indptr, target, side="left"))
        b = min(max(b, bounds[-1]), nrows)
        bounds.append(b)
    bounds.append(nrows)
    return bounds


class BlockCyclic:
    """Ownership of ``nrows`` rows by ``world`` ranks in ``blocks`` slabs.

    Slab j = padded rows [j*world*bs, (j+1)*world*bs); inside it rank r owns
    [(j*world + r)*bs, (j*world + r + 1)*bs).  ``padded`` >= nrows rows of
    storage make every block the same size, which is what an in-place
    all-gather needs; rows past ``nrows`` are never solved or read.
    """

    def __init__(self, nrows: int, world: int, blocks: int = 1, align: int = 16):
        self.nrows, self.world = int(nrows), int(world)
        self.blocks = max(1, int(blocks))
        per = -(-self.nrows // (self.world * self.blocks))
        self.bs = max(align, -(-per // align) * align)
        self.padded = self.bs * self.world * self.blocks

    def slab(self, j: int) -> Tuple[int, int]:
        return j * self.world * self.bs, (j + 1) * self.world * self.bs

    def block(self, j: int, rank: int) -> Tuple[int, int]:
        """Padded row range of rank's block in slab j."""
        a = (j * self.world + rank) * self.bs
        return a, a + self.bs

    def rows(self, j: int, rank: int) -> Tuple[int, int]:
        """Real rows of that block (clipped to nrows; may be empty)."""
        a, b = self.block(j, rank)
        return min(a, self.nrows), min(b, self.nrows)

    def owned(self, rank: int) -> List[Tuple[int, int]]:
        return [self.rows(j, rank) for j in range(self.blocks)]


def allgather_rows(T: torch.Tensor, bounds: Sequence[int], rank: int, group=None, scratch: Optional[dict] = None):
    """Make rows [bounds[r], bounds[r+1]) of T on every rank equal to rank r's
    copy.  Blocks differ in length, so each is sent padded to the longest."""
    world = len(bounds) - 1
    if world == 1:
        return
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    mx = max(max(sizes), 1)
    f = T.shape[1]
    key = (mx, f, T.dtype, T.device)
    if scratch is not None and scratch.get("key") == key:
        send, recv = scratch["send"], scratch["recv"]
    else:
        send = torch.zeros((mx, f), dtype=T.dtype, device=T.device)
        recv = torch.empty((world * mx, f), dtype=T.dtype, device=T.device)
        if scratch is not None:
            scratch.update(key=key, send=send, recv=recv)
    send[:sizes[rank]].copy_(T[bounds[rank]:bounds[rank + 1]])
    all_gather_list(list(recv.chunk(world, dim=0)), send, group=group)
    for r in range(world):
        if r != rank and sizes[r]:
            T[bounds[r]:bounds[r + 1]].copy_(recv[r * mx:r * mx + sizes[r]])


def _is_nccl(group=None) -> bool:
    return dist.get_backend(group) == "nccl"


class _Done:
    def wait(self):
        return True


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """``dist.all_gather_into_tensor``; device tensors on a backend that cannot take them (gloo: ranks that share
    one card, where RCCL refuses to run) go through host memory, synchronously."""
    if out.is_cuda and not _is_nccl(group):
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, inp.detach().cpu(), group=group)
        out.copy_(host_out)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


def all_gather_list(outs, inp: torch.Tensor, group=None):
    """``dist.all_gather`` with the same host staging."""
    if inp.is_cuda and not _is_nccl(group):
        host = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(host, inp.detach().cpu(), group=group)
        for o, h in zip(outs, host):
            o.copy_(h)
        return
    dist.all_gather(outs, inp, group=group)


def allreduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum over the ranks (through host memory when the backend cannot take the tensor's device)."""
    if dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and not _is_nccl(group):
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


def allreduce_max(t: torch.Tensor, group=None) -> torch.Tensor:
    """Maximum over the ranks (through host memory when the backend cannot take the tensor's device)."""
    if dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and not _is_nccl(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def alltoallv(send: Sequence[torch.Tensor], group=None) -> List[torch.Tensor]:
    """Rank r's ``send[d]`` (1-D, any length) arrives as element r of the list rank d gets back.  One
    all-to-all on RCCL; pairwise sends on other backends (through host memory for device tensors)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(send) == world
    dev, dtype = send[0].device, send[0].dtype
    nccl = _is_nccl(group)
    sizes = torch.tensor([int(t.numel()) for t in send], dtype=torch.int64, device=dev if nccl else "cpu")
    gathered = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(gathered, sizes, group=group)
    recv_sizes = [int(g[rank]) for g in gathered]
    if nccl:
        out = torch.empty(sum(recv_sizes), dtype=dtype, device=dev)
        dist.all_to_all_single(out, torch.cat(list(send)), recv_sizes, [int(t.numel()) for t in send], group=group)
        return list(out.split(recv_sizes))
    host = [t.detach().cpu().contiguous() for t in send]
    recv = [torch.empty(n, dtype=dtype) for n in recv_sizes]
    recv[rank] = host[rank]
    reqs = []
    for off in range(1, world):
        d, src = (rank + off) % world, (rank - off) % world
        if host[d].numel():
            reqs.append(dist.isend(host[d], d, group=group))
        if recv_sizes[src]:
            reqs.append(dist.irecv(recv[src], src, group=group))
    for r in reqs:
        r.wait()
    return [t.to(dev) for t in recv]


def shard_item_rows(blocks, users: "BlockCyclic", n_items: int, rank: int, group=None, transpose=None):
    """From the user-row shards to the item-row shard of the transposed matrix.

    ``blocks``: this rank's ``[(a, b, csr)]`` in slab order -- ``csr`` holds rows [a, b) of the interaction
    matrix (an ``engine.DeviceCSR`` or anything with ``indptr`` / ``indices`` / ``data`` tensors and ``shape``).
    Steps: per-item entry counts are summed over the ranks and cut into ``world`` contiguous item ranges of
    near-equal stored entries; every rank sends each other rank the entries of its rows that fall into that
    rank's item range (one all-to-all of row counts, column indices and values); the receiver strings the
    pieces together in user order -- a CSR of all users over its own item range -- and transposes that
    (``transpose(csr) -> csr``, by default the device transpose).  Rows of the result list users in ascending
    order with the values the whole matrix's transpose has, so the solves see the same numbers in the same
    order whatever the world size.  Returns ``(Ciu_local, item_bounds)``; nothing here holds more than this
    rank's share of the entries (plus one row-pointer array over all users)."""
    world = users.world
    assert [(a, b) for a, b, _ in blocks] == users.owned(rank), "blocks must be this rank's, in slab order"
    dev = blocks[0][2].indptr.device if blocks else torch.device("cpu")
    counts = torch.zeros(n_items, dtype=torch.int64, device=dev)
    for _, _, c in blocks:
        if c.indices.numel():
            counts += torch.bincount(c.indices, minlength=n_items)
    if world > 1:
        counts = allreduce_sum(counts, group)
    item_ptr = torch.zeros(n_items + 1, dtype=torch.int64, device=dev)
    item_ptr[1:] = torch.cumsum(counts, 0)
    bounds = partition_rows(item_ptr, world)
    if transpose is None:
        transpose = lambda m: m.transpose()  # noqa: E731
    from .engine import DeviceCSR

    if world == 1:
        if len(blocks) == 1:
            whole = blocks[0][2]
        else:   # one rank that holds its rows as several blocks: string them together, in row order
            ptrs, base = [torch.zeros(1, dtype=torch.int64, device=dev)], 0
            for _, _, c in blocks:
                p = c.indptr - c.indptr[0]
                ptrs.append(p[1:] + base)
                base += int(p[-1])
            whole = DeviceCSR(torch.cat(ptrs), torch.cat([c.indices for _, _, c in blocks]),
                              torch.cat([c.data for _, _, c in blocks]), (users.nrows, n_items))
        return transpose(whole), bounds

    # ---- split this rank's entries by destination item range
    cuts = torch.tensor(bounds[1:-1], dtype=torch.int32, device=dev)
    send = [[] for _ in range(world)]          # per destination: [row counts | columns | value bits] per block
    for a, b, c in blocks:
        nrows = b - a
        ptr = c.indptr - c.indptr[0]
        row_of = torch.repeat_interleave(torch.arange(nrows, dtype=torch.int32, device=dev), (ptr[1:] - ptr[:-1]))
        dest = torch.bucketize(c.indices, cuts, right=True)
        for d in range(world):
            sel = (dest == d).nonzero(as_tuple=True)[0]       # ascending: row-major order is kept
            rc = torch.bincount(row_of[sel], minlength=nrows).to(torch.int32) if nrows else row_of[:0]
            send[d].append((rc, (c.indices[sel] - bounds[d]).to(torch.int32), c.data[sel].view(torch.int32)))
        del row_of, dest
    packed = []
    for d in range(world):
        parts = [p[0] for p in send[d]] + [p[1] for p in send[d]] + [p[2] for p in send[d]]
        packed.append(torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int32, device=dev))
    del send
    got = alltoallv(packed, group)
    del packed

    # ---- string the pieces together in user order: CSR [all users] x [my item range]
    lo, hi = bounds[rank], bounds[rank + 1]
    cnt = torch.zeros(users.nrows, dtype=torch.int64, device=dev)
    pieces = []   # (first user, rows, columns, values) of every block of every source
    for src in range(world):
        owned = users.owned(src)
        nloc = sum(b - a for a, b in owned)
        buf = got[src]
        ne = (buf.numel() - nloc) // 2
        rc, cols, vals = buf[:nloc], buf[nloc:nloc + ne], buf[nloc + ne:].view(torch.float32)
        r0 = 0
        e_ends = torch.cumsum(rc.to(torch.int64), 0)
        for a, b in owned:
            if b > a:
                cnt[a:b] = rc[r0:r0 + (b - a)]
                e0 = int(e_ends[r0 - 1]) if r0 else 0
                e1 = int(e_ends[r0 + (b - a) - 1])
                pieces.append((a, cols[e0:e1], vals[e0:e1]))
                r0 += b - a
    indptr = torch.zeros(users.nrows + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(cnt, 0)
    total = int(indptr[-1])
    indices = torch.empty(total, dtype=torch.int32, device=dev)
    data = torch.empty(total, dtype=torch.float32, device=dev)
    starts = indptr[torch.tensor([p[0] for p in pieces], dtype=torch.int64, device=dev)].tolist() if pieces else []
    for (a, cols, vals), s0 in zip(pieces, starts):
        indices[s0:s0 + cols.numel()] = cols
        data[s0:s0 + vals.numel()] = vals
    del got, pieces
    return transpose(DeviceCSR(indptr, indices, data, (users.nrows, hi - lo))), bounds


class ShardedAlsDriver:
    """iteration() = for every slab: local user solve of this rank's block, start its in-place all-gather;
    wait for the gathers; local item solve; all-gather Y.

    ``Xpad`` has ``users.padded`` rows (``Xpad[:users.nrows]`` is X).  ``solve_users(a, b)`` /
    ``solve_items(a, b)`` solve rows [a, b) in place (a == b: nothing to do).
    """

    def __init__(self, Xpad: torch.Tensor, Y: torch.Tensor, users: BlockCyclic, item_bounds: Sequence[int],
                 solve_users: Callable[[int, int], None], solve_items: Callable[[int, int], None], rank: int,
                 group=None):
        assert Xpad.shape[0] == users.padded and Xpad.is_contiguous()
        if users.world > 1:
            # the in-place gather puts the piece of group rank r at block r of a slab: the two must be the same number
            assert dist.get_rank(group) == rank and dist.get_world_size(group) == users.world, \
                "the rank that owns block r of every slab must be rank r of the process group"
        self.Xpad, self.Y = Xpad, Y
        self.X = Xpad[:users.nrows]
        self.users, self.item_bounds = users, list(item_bounds)
        self.solve_users, self.solve_items = solve_users, solve_items
        self.rank, self.group = rank, group
        self._sy = {}

    @property
    def user_blocks(self) -> List[Tuple[int, int]]:
        return self.users.owned(self.rank)

    def iteration(self) -> None:
        r, u = self.rank, self.users
        pending = []
        for j in range(u.blocks):
            a, b = u.rows(j, r)
            if b > a:
                self.solve_users(a, b)
            if u.world > 1:
                s0, s1 = u.slab(j)
                p0, p1 = u.block(j, r)
                # async: the collective waits for the solve just enqueued and runs beside the next block's
                pending.append(all_gather_into(self.Xpad[s0:s1], self.Xpad[p0:p1], group=self.group, async_op=True))
        t0 = self._mark()
        for w in pending:
            w.wait()
        t1 = self._mark()
        self.solve_items(self.item_bounds[r], self.item_bounds[r + 1])
        t2 = self._mark()
        allgather_rows(self.Y, self.item_bounds, r, self.group, self._sy)
        if t0 is not None:
            self.events.append(("wait_user_exchange", t0, t1))
            self.events.append(("item_exchange", t2, self._mark()))

    # events = [] switches on the timing of what the compute stream spends on the exchanges (device tensors only):
    # the wait for the user slabs' gathers that the solves did not hide, and the item exchange
    events = None

    def _mark(self):
        if self.events is None or not self.Xpad.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def exchange_times_ms(self) -> dict:
        """Sum and count of the recorded intervals per name (call after a synchronize)."""
        out = {}
        for name, a, b in self.events or []:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + a.elapsed_time(b), n + 1)
        return out


def gpu_driver(Cui, Ciu, X: torch.Tensor, Y: torch.Tensor, regularization: float, cg_steps: int, rank: int,
               world: int, group=None, chunk_nnz: int = 0, short_max=None, blocks: Optional[int] = None):
    """Build the libmirec engine of this rank's shard and the driver around it.

    ``Cui``: either the whole interaction matrix (an ``engine.DeviceCSR``; sliced and re-distributed here -- for
    callers that have it anyway) or this rank's user-row shards ``[(a, b, csr)]`` for the blocks
    ``user_layout(n_users, world, blocks).owned(rank)``; ``Ciu`` is then ignored and the item-row shard is built
    by ``shard_item_rows``.  The driver's ``X`` (a view of padded storage holding a copy of the given X) is the
    user factor matrix from then on."""
    from . import engine as E

    n_users, n_items = int(X.shape[0]), int(Y.shape[0])
    users = user_layout(n_users, world, blocks)
    if isinstance(Cui, E.DeviceCSR):
        assert Cui.shape == (n_users, n_items)
        shards = [(a, b, Cui if (a, b) == (0, n_users) else Cui.row_slice(a, b)) for a, b in users.owned(rank)]
        if world == 1 and Ciu is not None:
            ciu_local, ib = Ciu, [0, n_items]
        else:
            ciu_local, ib = shard_item_rows(shards, users, n_items, rank, group)
    else:
        shards = list(Cui)
        ciu_local, ib = shard_item_rows(shards, users, n_items, rank, group)
    Xpad = torch.zeros((users.padded, X.shape[1]), dtype=X.dtype, device=X.device)
    Xpad[:users.nrows].copy_(X)
    # the largest |confidence| of the whole matrix: every rank scales the matrix-core operands alike
    cmax = torch.zeros(1, dtype=torch.float32, device=X.device)
    for _, _, c in shards:
        if c.nnz:
            cmax = torch.maximum(cmax, c.data.abs().max().reshape(1))
    if world > 1:
        cmax = allreduce_max(cmax, group)
    eng = E.AlsEngine(None, None, Xpad[:users.nrows], Y, regularization, cg_steps, user_blocks=users.owned(rank),
                      item_rows=(ib[rank], ib[rank + 1]), chunk_nnz=chunk_nnz, short_max=short_max,
                      user_shards=[c for _, _, c in shards], item_shard=ciu_local, conf_max=float(cmax.item()))
    # The driver walks this rank's blocks in order and skips those without rows; block 0 (which refreshes the
    # Gram of Y) is empty only if all of the rank's blocks are, since later slabs hold higher rows.
    index_of = {blk: j for j, blk in enumerate(eng.user_blocks) if blk[1] > blk[0]}

    def solve_users(a, b):
        eng.user_sweep_block(index_of[(a, b)])

    drv = ShardedAlsDriver(Xpad, Y, users, ib, solve_users, lambda a, b: eng.item_sweep(), rank, group)
    return drv, eng


def user_layout(n_users: int, world: int, blocks: Optional[int] = None) -> BlockCyclic:
    """The block-cyclic ownership of the users that ``gpu_driver`` uses (one block on one rank, four slabs
    otherwise, so that three of the four exchanges of user factors hide behind the next block's solve)."""
    if blocks is None:
        blocks = 1 if world == 1 else 4
    return BlockCyclic(n_users, world, blocks)
