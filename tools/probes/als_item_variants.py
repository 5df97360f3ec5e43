"""Time the explicit-Gram item half-sweep of one build of the library and print a digest of its result.

usage: als_item_variants.py <path to libmirec.so> [users items mean_degree]
ITEM_CHUNK=<entries> in the environment overrides the chunk length of the plan (0 / unset: chosen from the matrix);
the time then includes the finalize kernel, since the number of split rows moves with it.
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
f = 128
dev = torch.device("cuda", 0)
indptr, indices, data = synth.interactions_torch(42, U, I, deg, dev)
Ciu = E.DeviceCSR(indptr, indices, data * 40.0, (U, I)).transpose()
g = torch.Generator(device=dev)
g.manual_seed(1)
X = torch.rand((U, f), device=dev, generator=g) * 0.1
Y0 = torch.rand((I, f), device=dev, generator=g) * 0.1
G = E.gram(X, 0.01)
import os  # noqa: E402
chunk = int(os.environ.get("ITEM_CHUNK", "0"))
plan = E.SweepPlan(Ciu, f, chunk)
phase = "main" if chunk == 0 else "both"
Y = Y0.clone()
plan.sweep(Y, X, G, 3, phase)
torch.cuda.synchronize()
digest = hashlib.sha256(Y.cpu().numpy().tobytes()).hexdigest()[:16]
reps = 5
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    plan.sweep(Y, X, G, 3, phase)
b.record()
torch.cuda.synchronize()
print("%s  chunk length %d  chunks %d  split rows %d  %.3f ms (%s)  digest %s" % (
    sys.argv[1].rsplit("/", 1)[-1], plan.chunk_nnz, plan.n_chunks, plan.n_split_rows, a.elapsed_time(b) / reps,
    "explicit-Gram kernel" if phase == "main" else "explicit-Gram kernel + finalize", digest), flush=True)
