"""Phase timing of the bf16-plane top-K kernel: needs a library whose F = 128 unit was built with the cycle-counter
patch (tools/probes/patches) and exports mirec_topk_debug_read.

usage: topk_phase_timing.py <libmirec variant> [ALS iterations]
Prints, per tile and per wavefront, the mean cycle counts of the sections of the sweep: with random factors, or --
with a number of iterations -- with the benchmark's factors trained that long, the items in their own order and by
descending interactions.
"""
import ctypes
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
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if iters:
    import numpy as np
    rs = np.random.RandomState(42)
    X = E.pad_factors(rs.rand(U, F).astype(np.float32) * 0.01, F, dev)
    Y = E.pad_factors(rs.rand(I, F).astype(np.float32) * 0.01, F, dev)
    eng = E.AlsEngine(E.DeviceCSR(indptr, indices, data * 40.0, (U, I)), None, X, Y, 0.01, 3)
    for _ in range(iters):
        eng.iteration()
    torch.cuda.synchronize()
else:
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    X = torch.randn((U, F), device=dev, generator=g) * 0.3
    Y = torch.randn((I, F), device=dev, generator=g) * 0.3
ordered = E.DeviceCSR(indptr, indices, data, (U, I))
ordered.build_order()
lib = _lib.require()
out = (ctypes.c_ulonglong * 16)()
ntile = (I + 31) // 32
for mask in (Cui, ordered, None):
    E.topk(X, Y, 20, mask)
    torch.cuda.synchronize()
    lib.mirec_topk_debug_read(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    E.topk(X, Y, 20, mask)
    e1.record()
    torch.cuda.synchronize()
    lib.mirec_topk_debug_read(out, 1)
    v = list(out)
    nw = max(v[6], 1)           # wavefronts
    print("filter=%s ordered=%s %.2f ms" % (mask is not None, mask is ordered, e0.elapsed_time(e1)))
    print("  cycles per wavefront: prologue %.0f, tiles swept %.0f (%.0f per tile), marked tiles worked %.0f "
          "(%.1f tiles, %.0f each), compactions at stretch ends %.0f, final stage %.0f"
          % (v[0] / nw, v[1] / nw, v[1] / nw / ntile, v[2] / nw, v[5] / nw, v[2] / max(v[5], 1), v[3] / nw, v[4] / nw),
          flush=True)
    # (the sweep as every wavefront sees it, the wait at the barriers included; inside the sweeping half:)
    print("  sweeping wavefronts alone: %.0f cycles per tile, %.0f of them in the loop over the tiles"
          % (v[7] / (nw / 2) / ntile, v[8] / (nw / 2) / ntile), flush=True)
