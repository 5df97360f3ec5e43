"""Phase timing of the matrix-free half-sweep (library built with -DMIREC_DEBUG_TIMING for als_short.hip).

usage: short_phase_timing.py <libmirec variant>
Prints the mean cycle counts per row of the phases of the first solver of every workgroup.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import _lib

_lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

U, I, deg, f = 1_000_000, 27_000, 50.0, 128
dev = torch.device("cuda", 0)
indptr, indices, data = synth.interactions_torch(42, U, I, deg, dev)
Cui = E.DeviceCSR(indptr, indices, data * 40.0, (U, I))
rs = np.random.RandomState(42)
X = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, f, dev)
Y = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, f, dev)
G = E.gram(Y, 0.01)
plan = E.SweepPlan(Cui, f)
lib = _lib.require()
lib.mirec_short_debug_read.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 16)()
plan.sweep(X, Y, G, 3, "short")
torch.cuda.synchronize()
lib.mirec_short_debug_read(out, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
plan.sweep(X, Y, G, 3, "short")
b.record()
torch.cuda.synchronize()
lib.mirec_short_debug_read(out, 1)
v = list(out)
rows = max(v[7], 1)
names = ["row header + x + norm", "publish (4 per row)", "first pass", "collect waits (4 per row)", "later passes (3 per row)",
         "vector updates (3 per row)", "whole row"]
print("%.3f ms; %d rows timed; cycles per row:" % (a.elapsed_time(b), rows), "; ".join("%s %.0f" % (n, v[i] / rows) for i, n in enumerate(names)))
