"""The collectives of the sharded driver on the RCCL backend with a single rank (all that one GPU allows):
checks that RCCL initialises on this box and takes the calls as the driver makes them -- the variable-size
all-to-all of int32 payloads, the all-gather of sizes, the in-place all-gather of a row block into its slab,
the padded all-gather of item rows, the maximum / sum reductions.  Prints "rccl one-rank collectives: ok".

Run under torch.distributed.run with one process (tests/test_gpu_als.py does).
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from movie_recommender_system_amd import dist as D  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1 and D._is_nccl()

# alltoallv: one destination, odd length
payload = torch.arange(12345, dtype=torch.int32, device=dev)
got = D.alltoallv([payload])
assert len(got) == 1 and torch.equal(got[0], payload)
got = D.alltoallv([payload[:0]])
assert got[0].numel() == 0

# the in-place all-gather of a block into its slab, asynchronous as in ShardedAlsDriver.iteration
X = torch.randn((4096, 128), device=dev)
ref = X.clone()
w = dist.all_gather_into_tensor(X[1024:2048], X[1024:2048], async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(X, ref)

# reductions as the driver uses them (the helpers return at once for one rank: call the backend directly)
t = torch.tensor([3.5], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.arange(7, dtype=torch.int64, device=dev)
dist.all_reduce(c)
assert float(t) == 3.5 and torch.equal(c, torch.arange(7, dtype=torch.int64, device=dev))

# padded all-gather of item rows (list form)
Y = torch.randn((100, 128), device=dev)
recv = torch.empty_like(Y)
dist.all_gather([recv], Y)
assert torch.equal(recv, Y)
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank collectives: ok")
