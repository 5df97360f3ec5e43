"""Rehearsal of bench.py's multi-rank path on ONE GPU.

RCCL refuses two ranks on one device, so the ranks talk over gloo and the two
collectives of the data path (in-place all-gather of user slabs, padded
all-gather of item blocks) are staged through host memory here -- slow, and
only meant to exercise the sharding, the block-cyclic solves, the per-rank
top-K and the JSON line before the real 2/4/8-GPU runs.  Launch:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/rehearse_multi_rank.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIREC_BENCH_BACKEND"] = "gloo"
os.environ["MIREC_BENCH_DEVICE"] = "0"


class _Done:
    def wait(self):
        return True


_ag_into, _ag = dist.all_gather_into_tensor, dist.all_gather


def all_gather_into_tensor(out, inp, group=None, async_op=False):
    if not out.is_cuda:
        return _ag_into(out, inp, group=group, async_op=async_op)
    host_in = inp.detach().cpu()
    host_out = torch.empty(out.shape, dtype=out.dtype)
    _ag_into(host_out, host_in, group=group)
    out.copy_(host_out)
    return _Done()


def all_gather(outs, inp, group=None, async_op=False):
    if not inp.is_cuda:
        return _ag(outs, inp, group=group, async_op=async_op)
    host = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
    _ag(host, inp.detach().cpu(), group=group)
    for o, h in zip(outs, host):
        o.copy_(h)
    return _Done()


dist.all_gather_into_tensor = all_gather_into_tensor
dist.all_gather = all_gather

import bench  # noqa: E402

bench.main()
