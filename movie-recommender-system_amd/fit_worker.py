"""One rank of a multi-GPU ``AlternatingLeastSquares.fit``.

``fit`` with ``devices=[d0, d1, ...]`` starts one fresh Python process per entry, each running this module
before anything in it has touched a GPU; the ranks shard users and items as ``dist.gpu_driver`` does (user
blocks block-cyclic with in-place all-gathers, item ranges by stored entries, interaction rows stored by owner
only) and talk over RCCL -- or over gloo through host memory when two ranks name the same card, which RCCL
refuses (a rehearsal mode: the sharding is the same, the exchange is slow).  Inputs come from a scratch
directory the parent wrote (CSR arrays, start factors, parameters; memory-mapped here, so a rank pages in only
the rows it owns); rank 0 writes the fitted factors back.

    python -m movie_recommender_system_amd.fit_worker <scratch dir> <rank>
"""

from __future__ import annotations

import json
import os
import sys


def main(argv) -> int:
    scratch, rank = argv[1], int(argv[2])
    with open(os.path.join(scratch, "job.json")) as fh:
        job = json.load(fh)
    world = len(job["devices"])
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(job["port"]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    from . import dist as D
    from . import engine as E

    index = int(job["devices"][rank])
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if job["backend"] == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")

    def mm(name):
        return np.load(os.path.join(scratch, name + ".npy"), mmap_mode="r")

    indptr, indices, data = mm("indptr"), mm("indices"), mm("data")
    U, I, f = int(job["users"]), int(job["items"]), int(job["factors"])
    F = E.factor_width(f)
    X = E.pad_factors(np.asarray(mm("x0")), F, dev)
    Y = E.pad_factors(np.asarray(mm("y0")), F, dev)
    scale = np.float32(job["alpha"])
    shards = []
    for a, b in D.user_layout(U, world).owned(rank):
        e0, e1 = int(indptr[a]), int(indptr[b])
        vals = np.ascontiguousarray(data[e0:e1], dtype=np.float32)
        if scale != 1.0:
            vals = vals * scale
        shards.append((a, b, E.DeviceCSR(
            torch.from_numpy(np.ascontiguousarray(indptr[a:b + 1], dtype=np.int64) - e0).to(dev),
            torch.from_numpy(np.ascontiguousarray(indices[e0:e1], dtype=np.int32)).to(dev),
            torch.from_numpy(vals).to(dev), (b - a, I))))
    exact = not job["use_cg"]
    drv, _eng = D.gpu_driver(shards, None, X, Y, float(job["regularization"]),
                             E_SOLVE_EXACT if exact else int(job["cg_steps"]), rank, world,
                             chunk_nnz=int(job["chunk_nnz"]), short_max=0 if exact else None)
    for _ in range(int(job["iterations"])):
        drv.iteration()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(os.path.join(scratch, "x_out.npy"), drv.X[:, :f].contiguous().cpu().numpy())
        np.save(os.path.join(scratch, "y_out.npy"), drv.Y[:, :f].contiguous().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    return 0


E_SOLVE_EXACT = -1   # MIREC_SOLVE_EXACT of the C ABI

if __name__ == "__main__":
    sys.exit(main(sys.argv))
