"""Two ranks on one GPU (gloo, host-staged collectives as in tools/rehearse_multi_rank.py): the factors after two
sharded iterations must equal, bit for bit, those of a one-rank run of the same problem.  Launch with
torch.distributed.run --nproc-per-node 2.  REHEARSE_USERS / REHEARSE_ITEMS override the problem size (small ones
leave ranks and blocks without rows); REHEARSE_EXACT=1 runs the exact (Cholesky) solver instead of the CG steps."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movie_recommender_system_amd import dist as D, engine as E, synth  # noqa: E402


class _Done:
    def wait(self):
        return True


_ag_into, _ag = dist.all_gather_into_tensor, dist.all_gather


def _into(out, inp, group=None, async_op=False):
    host_out = torch.empty(out.shape, dtype=out.dtype)
    _ag_into(host_out, inp.detach().cpu(), group=group)
    out.copy_(host_out)
    return _Done()


def _list(outs, inp, group=None, async_op=False):
    host = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
    _ag(host, inp.detach().cpu(), group=group)
    for o, h in zip(outs, host):
        o.copy_(h)
    return _Done()


dist.all_gather_into_tensor, dist.all_gather = _into, _list
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
U, I, f = int(os.environ.get("REHEARSE_USERS", 60_000)), int(os.environ.get("REHEARSE_ITEMS", 5_000)), 128
rs = np.random.RandomState(7)
EXACT = os.environ.get("REHEARSE_EXACT", "0") == "1"
CG, SHORT_MAX = (-1, 0) if EXACT else (3, None)
X0 = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, f, dev)
Y0 = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, f, dev)


def run(r, w):
    # every rank generates the rows of its own user blocks only; the item-row shard comes from the exchange
    shards = []
    for a, b in D.user_layout(U, w).owned(r):
        ip, ix, dt = synth.interactions_torch(7, U, I, 40.0, dev, users=(a, b))
        shards.append((a, b, E.DeviceCSR(ip, ix, dt * 40.0, (b - a, I))))
    if w == 1:
        drv, eng = D.gpu_driver(shards[0][2], None, X0.clone(), Y0.clone(), 0.01, CG, 0, 1, short_max=SHORT_MAX)
    else:
        drv, eng = D.gpu_driver(shards, None, X0.clone(), Y0.clone(), 0.01, CG, r, w, short_max=SHORT_MAX)
    for _ in range(2):
        drv.iteration()
    torch.cuda.synchronize()
    return drv.X.clone(), drv.Y.clone(), eng.resident_matrix_bytes()


Xs, Ys, held = run(rank, world)
if rank == 0:
    X1, Y1, whole = run(0, 1)
    print(f"interaction-matrix bytes on rank 0: {held} of {whole} ({held / whole:.2f})")
    ok = torch.equal(Xs, X1) and torch.equal(Ys, Y1)
    print("world", world, "vs 1: factors bit-identical:", ok, "| max abs diff X", float((Xs - X1).abs().max()), "Y",
          float((Ys - Y1).abs().max()))
    if not ok:
        sys.exit(1)
dist.barrier()
