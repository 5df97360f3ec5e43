"""Time the matrix-free user half-sweep of one build of the library and print a digest of its result.

usage: als_short_variants.py <path to libmirec.so> [users items mean_degree [width]]
Builds of the kernel that differ only in how rows are mapped to wavefronts and where gathered rows are kept
must print the same digest: the arithmetic of a row does not depend on either.
"""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from movie_recommender_system_amd import _lib

_lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

U, I, deg = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1_000_000, 27_000, 50.0)
f = int(sys.argv[5]) if len(sys.argv) > 5 else 128
dev = torch.device("cuda", 0)
indptr, indices, data = synth.interactions_torch(42, U, I, deg, dev)
Cui = E.DeviceCSR(indptr, indices, data * 40.0, (U, I))
rs = np.random.RandomState(42)
X0 = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, f, dev)
Y = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, f, dev)
G = E.gram(Y, 0.01)
plan = E.SweepPlan(Cui, f)
X = X0.clone()
plan.sweep(X, Y, G, 3, "short")
torch.cuda.synchronize()
digest = hashlib.sha256(X.cpu().numpy().tobytes()).hexdigest()[:16]
reps = 5
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    plan.sweep(X, Y, G, 3, "short")
b.record()
torch.cuda.synchronize()
print("%s  short_max %d  short rows %d  cache rows %d  %.3f ms  digest %s" % (
    sys.argv[1].rsplit("/", 1)[-1], plan.short_max, plan.n_short_rows,
    int(_lib.require().mirec_als_short_cache_rows(f)), a.elapsed_time(b) / reps, digest), flush=True)
