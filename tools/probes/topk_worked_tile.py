"""Where the work on a marked tile of the top-K sweep goes: needs a library whose F = 128 unit was built with the
worked-tile cycle stamps (tools/probes/build_variant.py <name> topk_inst_3.hip -DMIREC_DEBUG_TIMING_WORKED).

usage: topk_worked_tile.py <libmirec variant>
The benchmark's factors after six iterations, items by descending interactions.
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import _lib
_lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth
dev = torch.device("cuda", 0)
U, I, F = 1_000_000, 27_000, 128
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
rs = np.random.RandomState(42)
X = E.pad_factors(rs.rand(U, F).astype(np.float32) * 0.01, F, dev)
Y = E.pad_factors(rs.rand(I, F).astype(np.float32) * 0.01, F, dev)
eng = E.AlsEngine(E.DeviceCSR(indptr, indices, data * 40.0, (U, I)), None, X, Y, 0.01, 3)
for _ in range(6): eng.iteration()
m = E.DeviceCSR(indptr, indices, data, (U, I)); m.build_order()
lib = _lib.require(); out = (ctypes.c_ulonglong * 16)()
E.topk(X, Y, 20, m); torch.cuda.synchronize(); lib.mirec_topk_debug_read(out, 1)
E.topk(X, Y, 20, m); torch.cuda.synchronize(); lib.mirec_topk_debug_read(out, 1)
v = list(out); n = max(v[5], 1)
print("worked tiles per wavefront %.1f; cycles per worked tile: operand loads %.0f, products %.0f, liked bits %.0f, selection %.0f, re-offers %.0f"
      % (v[5] / v[6], v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n))
