"""The sharded driver on CPU: gloo ranks that each generate only the rows of their
own user blocks, exchange them into item-row shards (``shard_item_rows``) and
solve with the oracle give bit-for-bit the factors of the single-process oracle
fit (row solves are independent given the gathered factors, and every rank
recomputes the Gram from identical bits)."""

import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from movie_recommender_system_amd import dist as D  # noqa: E402
from movie_recommender_system_amd import synth  # noqa: E402


def test_partition_rows_balances_entries():
    indptr = np.array([0, 10, 10, 11, 50, 51, 100], dtype=np.int64)
    b = D.partition_rows(indptr, 2)
    assert b[0] == 0 and b[-1] == 6 and b == sorted(b)
    left = indptr[b[1]] - indptr[0]
    assert abs(left - 50) <= 40
    assert D.partition_rows(indptr, 1) == [0, 6]
    b4 = D.partition_rows(torch.from_numpy(indptr), 4)
    assert len(b4) == 5 and b4 == sorted(b4)
    # more ranks than rows: empty blocks are allowed
    b8 = D.partition_rows(np.array([0, 3, 5], dtype=np.int64), 8)
    assert len(b8) == 9 and b8[-1] == 2 and b8 == sorted(b8)
    # a shard's pointer array does not start at 0
    assert D.partition_rows(np.array([100, 110, 120, 130, 140], dtype=np.int64), 2) == [0, 2, 4]


def test_block_cyclic_ownership():
    u = D.BlockCyclic(1000, 4, blocks=3)
    assert u.padded >= 1000 and u.padded == u.bs * 12 and u.bs % 16 == 0
    seen = np.zeros(u.padded, dtype=np.int32)
    for r in range(4):
        owned = u.owned(r)
        assert len(owned) == 3
        for j, (a, b) in enumerate(owned):
            pa, pb = u.block(j, r)
            s0, s1 = u.slab(j)
            assert s0 <= pa < pb <= s1 and pb - pa == u.bs
            assert (a, b) == (min(pa, 1000), min(pb, 1000))
            seen[pa:pb] += 1
    assert np.all(seen == 1)                       # every padded row has exactly one owner
    one = D.BlockCyclic(10, 1)
    assert one.owned(0) == [(0, 10)] and one.padded >= 10
    tiny = D.BlockCyclic(5, 8, blocks=2)           # more blocks than rows: empty ranges are allowed
    assert sum(b - a for r in range(8) for a, b in tiny.owned(r)) == 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_transpose(m):
    """``.T.tocsr()`` through scipy: stands in for the device transpose on CPU tensors."""
    from movie_recommender_system_amd.engine import DeviceCSR
    T = sp.csr_matrix(m.to_scipy().T)
    T.sort_indices()
    return DeviceCSR.from_scipy(T, torch.device("cpu"))


def _worker(rank, world, port, out_dir, nu=300, ni=90, blocks=3, deg=10.0):
    import oracle
    from movie_recommender_system_amd.engine import DeviceCSR
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        users = D.BlockCyclic(nu, world, blocks=blocks)
        # every rank generates the rows of its own user blocks and nothing else ...
        shards = []
        for a, b in users.owned(rank):
            ip, ix, dt = synth.interactions_torch(3, nu, ni, deg, "cpu", users=(a, b))
            shards.append((a, b, DeviceCSR(ip, ix, dt * 40.0, (b - a, ni))))
        # ... and receives the rows of the transposed matrix for its item range
        ciu_local, ib = D.shard_item_rows(shards, users, ni, rank, transpose=_cpu_transpose)
        whole = sp.csr_matrix(synth.interactions(3, nu, ni, deg) * np.float32(40.0))
        want = sp.csr_matrix(whole.T)
        want.sort_indices()
        assert ib == D.partition_rows(want.indptr, world)
        want = want[ib[rank]:ib[rank + 1]]
        got = ciu_local.to_scipy()
        assert got.shape == want.shape and np.array_equal(got.indptr, want.indptr)
        assert np.array_equal(got.indices, want.indices) and np.array_equal(got.data, want.data)
        held = sum(c.indices.numel() for _, _, c in shards) + ciu_local.indices.numel()
        X0, Y0 = oracle.oracle.init_factors(nu, ni, 16, 42)
        Xpad = torch.zeros((users.padded, 16), dtype=torch.float32)
        Xpad[:nu] = torch.from_numpy(X0)
        Y = torch.from_numpy(Y0.copy())
        Xn, Yn = Xpad[:nu].numpy(), Y.numpy()   # views: the oracle updates the tensors in place
        local_u = {(a, b): c.to_scipy() for a, b, c in shards}
        local_i = ciu_local.to_scipy()
        drv = D.ShardedAlsDriver(
            Xpad, Y, users, ib,
            lambda a, b: oracle.als_half_sweep(local_u[(a, b)], Xn[a:b], Yn, 0.01, 3),
            lambda a, b: oracle.als_half_sweep(local_i, Yn[a:b], Xn, 0.01, 3),
            rank)
        for _ in range(3):
            drv.iteration()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), X=Xn, Y=Yn, ub=np.array(drv.user_blocks), ib=np.array(ib),
                 held=held, nnz=whole.nnz)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_single_process(tmp_path):
    import oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    M = synth.interactions(3, 300, 90, 10.0)
    Xr, Yr = oracle.als_fit(M, factors=16, regularization=0.01, alpha=40.0, iterations=3, random_state=42)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert r0["ub"].shape == (3, 2) and not np.array_equal(r0["ub"], r1["ub"]) and 0 < r0["ib"][1] < 90
    for r in (r0, r1):
        assert np.array_equal(r["X"], Xr)
        assert np.array_equal(r["Y"], Yr)
    # storage is sharded: the two ranks together hold every entry once per orientation
    assert int(r0["held"]) + int(r1["held"]) == 2 * int(r0["nnz"])
    assert 0.35 * 2 * int(r0["nnz"]) < int(r0["held"]) < 0.65 * 2 * int(r0["nnz"])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,nu,ni,blocks", [(3, 5, 2, 2), (4, 37, 9, 1), (3, 100, 40, 4)])
def test_odd_worlds_and_shards_without_rows(tmp_path, world, nu, ni, blocks):
    """More ranks than items, blocks without users, a world size that divides nothing: every rank still ends
    with the factors of the single-process fit, bit for bit."""
    import oracle
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), nu, ni, blocks, 3.0), nprocs=world, join=True)
    M = synth.interactions(3, nu, ni, 3.0)
    Xr, Yr = oracle.als_fit(M, factors=16, regularization=0.01, alpha=40.0, iterations=3, random_state=42)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["X"], Xr), r
        assert np.array_equal(got["Y"], Yr), r


def test_one_rank_with_several_blocks_strings_them_together():
    """A single rank may hold its user rows as several blocks (no process group exists then): the item-row
    shard is the transpose of the blocks strung together in row order."""
    from movie_recommender_system_amd.engine import DeviceCSR
    nu, ni = 200, 60
    users = D.BlockCyclic(nu, 1, blocks=3)
    shards = []
    for a, b in users.owned(0):
        ip, ix, dt = synth.interactions_torch(5, nu, ni, 6.0, "cpu", users=(a, b))
        shards.append((a, b, DeviceCSR(ip, ix, dt, (b - a, ni))))
    assert len(shards) == 3 and sum(b - a for a, b, _ in shards) == nu
    ciu, ib = D.shard_item_rows(shards, users, ni, 0, transpose=_cpu_transpose)
    want = sp.csr_matrix(synth.interactions(5, nu, ni, 6.0).T)
    want.sort_indices()
    got = ciu.to_scipy()
    assert ib == [0, ni] and got.shape == want.shape
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.array_equal(got.data, want.data)
