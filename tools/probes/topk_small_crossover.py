"""Where the per-request kernels (items split over the workgroups) hand over to the all-users kernels
(queries split): device time of one call of each for a growing number of queries, 27 k items, top-20 with
the liked-item filter; then the wall time of single requests through the raw library call.

usage: topk_small_crossover.py [f=128]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
U, I = 20_000, 27_000
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
Cui = E.DeviceCSR(indptr, indices, data, (U, I))
g = torch.Generator(device=dev)
g.manual_seed(1)
X = torch.randn((U, F), device=dev, generator=g) * 0.3
Y = torch.randn((I, F), device=dev, generator=g) * 0.3


def dev_ms(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


print(f"f={F} items={I} k=20, liked-item filter; median device ms per call")
for nq in (1, 2, 8, 32, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
    Q = X[:nq].contiguous()
    m = Cui.row_slice(0, nq)
    small = dev_ms(lambda: E.topk(Q, Y, 20, m, path="small"))
    sweep = dev_ms(lambda: E.topk(Q, Y, 20, m, path="sweep"))
    a, _ = E.topk(Q, Y, 20, m, path="small")
    b, _ = E.topk(Q, Y, 20, m, path="sweep")
    print(f"nq={nq:6d}  small {small:8.3f}  sweep {sweep:8.3f}  equal={bool(torch.equal(a, b))}", flush=True)

# wall time of a request: nothing uploaded, 160 bytes downloaded
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
out = (torch.empty((1, 20), dtype=torch.int32, device=dev), torch.empty((1, 20), dtype=torch.float32, device=dev))
ts = []
for u in range(2000):
    t0 = time.perf_counter()
    E.topk_small(X[u:u + 1], Y, 20, Cui.row_slice(u, u + 1), ws=ws, out=out)
    ids = out[0].cpu()
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[200:])
print(f"request wall time (row_slice + call + download): p50 {ts[len(ts) // 2] * 1e6:.0f} us  p99 {ts[int(len(ts) * 0.99)] * 1e6:.0f} us")
