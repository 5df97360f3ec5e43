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
    dist.all_gather(list(recv.chunk(world, dim=0)), send, group=group)
    for r in range(world):
        if r != rank and sizes[r]:
            T[bounds[r]:bounds[r + 1]].copy_(recv[r * mx:r * mx + sizes[r]])


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
                pending.append(dist.all_gather_into_tensor(self.Xpad[s0:s1], self.Xpad[p0:p1], group=self.group,
                                                           async_op=True))
        for w in pending:
            w.wait()
        self.solve_items(self.item_bounds[r], self.item_bounds[r + 1])
        allgather_rows(self.Y, self.item_bounds, r, self.group, self._sy)


def gpu_driver(Cui, Ciu, X: torch.Tensor, Y: torch.Tensor, regularization: float, cg_steps: int, rank: int,
               world: int, group=None, chunk_nnz: int = 0, short_max=None, blocks: Optional[int] = None):
    """Build the libmirec engine of this rank's shard and the driver around it.
    `Cui` / `Ciu` are whole `engine.DeviceCSR` matrices on this rank's GPU.  The
    driver's ``X`` (a view of padded storage holding a copy of the given X) is
    the user factor matrix from then on."""
    from . import engine as E

    if Ciu is None:
        Ciu = Cui.transpose()
    if blocks is None:
        blocks = 1 if world == 1 else 4
    users = BlockCyclic(Cui.shape[0], world, blocks)
    Xpad = torch.zeros((users.padded, X.shape[1]), dtype=X.dtype, device=X.device)
    Xpad[:users.nrows].copy_(X)
    ib = partition_rows(Ciu.indptr, world)
    eng = E.AlsEngine(Cui, Ciu, Xpad[:users.nrows], Y, regularization, cg_steps, user_blocks=users.owned(rank),
                      item_rows=(ib[rank], ib[rank + 1]), chunk_nnz=chunk_nnz, short_max=short_max)
    # The driver walks this rank's blocks in order and skips those without rows; block 0 (which refreshes the
    # Gram of Y) is empty only if all of the rank's blocks are, since later slabs hold higher rows.
    index_of = {blk: j for j, blk in enumerate(eng.user_blocks) if blk[1] > blk[0]}

    def solve_users(a, b):
        eng.user_sweep_block(index_of[(a, b)])

    drv = ShardedAlsDriver(Xpad, Y, users, ib, solve_users, lambda a, b: eng.item_sweep(), rank, group)
    return drv, eng
