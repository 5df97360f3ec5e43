"""Time of one all-users top-20 pass (1 M x 27 k, F = 128) with and without the liked-item filter.

usage: topk_time.py [libmirec.so]   (a variant built by tools/probes/build_variant.py, or the regular build)
"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

dev = torch.device("cuda", 0)
U, I, F = 1_000_000, 27_000, 128
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
Cui = E.DeviceCSR(indptr, indices, data, (U, I))
g = torch.Generator(device=dev)
g.manual_seed(1)
X = torch.randn((U, F), device=dev, generator=g) * 0.3
Y = torch.randn((I, F), device=dev, generator=g) * 0.3
name = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "libmirec.so"
for mask in ((Cui, None) if os.environ.get("TK_MASK", "1") == "1" else (None,)):
    ids, sc = E.topk(X, Y, 20, mask)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ids, sc = E.topk(X, Y, 20, mask)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    digest = int(ids.to(torch.int64).sum().item()) ^ int(sc.view(torch.int32).to(torch.int64).sum().item())
    print("%s  filter=%s  min %.2f ms  median %.2f ms  digest %016x" % (name, mask is not None, min(ts), sorted(ts)[2], digest & (2**64 - 1)))
