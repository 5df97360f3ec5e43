"""Row-sharded ALS over the GPUs of one node (one process per GPU).

Rows of a half-sweep are independent given the other side's factors and its
Gram, so users are split into contiguous blocks balanced by stored entries for
the user half-sweep, items likewise for the item half-sweep.  Every rank keeps
both factor matrices whole; after a half-sweep the freshly solved blocks are
exchanged with one all-gather (RCCL over xGMI when the process group's backend
is "nccl").  The Gram of the gathered matrix is recomputed locally by every
rank instead of all-reduced, so every rank -- and every world size -- works
from identical bits.  Scoring needs no exchange: each rank scores its users.

The driver takes the two local solves as callables, so the same partition and
exchange logic runs on CPU tensors over gloo in tests.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Sequence

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
    """iteration() = local user solve, all-gather X, local item solve, all-gather Y."""

    def __init__(self, X: torch.Tensor, Y: torch.Tensor, user_bounds: Sequence[int], item_bounds: Sequence[int],
                 solve_users: Callable[[int, int], None], solve_items: Callable[[int, int], None], rank: int,
                 group=None):
        self.X, self.Y = X, Y
        self.user_bounds, self.item_bounds = list(user_bounds), list(item_bounds)
        self.solve_users, self.solve_items = solve_users, solve_items
        self.rank, self.group = rank, group
        self._sx, self._sy = {}, {}

    def iteration(self) -> None:
        r = self.rank
        self.solve_users(self.user_bounds[r], self.user_bounds[r + 1])
        allgather_rows(self.X, self.user_bounds, r, self.group, self._sx)
        self.solve_items(self.item_bounds[r], self.item_bounds[r + 1])
        allgather_rows(self.Y, self.item_bounds, r, self.group, self._sy)


def gpu_driver(Cui, Ciu, X: torch.Tensor, Y: torch.Tensor, regularization: float, cg_steps: int, rank: int,
               world: int, group=None, chunk_nnz: int = 0, short_max=None):
    """Build the libmirec engine of this rank's shard and the driver around it.
    `Cui` / `Ciu` are whole `engine.DeviceCSR` matrices on this rank's GPU."""
    from . import engine as E

    if Ciu is None:
        Ciu = Cui.transpose()
    ub = partition_rows(Cui.indptr, world)
    ib = partition_rows(Ciu.indptr, world)
    eng = E.AlsEngine(Cui, Ciu, X, Y, regularization, cg_steps, user_rows=(ub[rank], ub[rank + 1]),
                      item_rows=(ib[rank], ib[rank + 1]), chunk_nnz=chunk_nnz, short_max=short_max)
    drv = ShardedAlsDriver(X, Y, ub, ib, lambda a, b: eng.user_sweep(), lambda a, b: eng.item_sweep(), rank, group)
    return drv, eng
