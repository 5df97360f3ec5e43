"""Time one ALS iteration with the exact (Cholesky) solver beside the CG-3 iteration on the benchmark matrix,
and report how far both are from the float64-carried exact solve of the oracle on sampled rows.

usage: exact_iteration.py [users items mean_degree factors]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

U, I, deg, f = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 \
    else (1_000_000, 27_000, 50.0, 128)
dev = torch.device("cuda", 0)
indptr, indices, data = synth.interactions_torch(42, U, I, deg, dev)
Cui = E.DeviceCSR(indptr, indices, data * 40.0, (U, I))
Ciu = Cui.transpose()
rs = np.random.RandomState(42)
F = E.factor_width(f)
X0 = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, F, dev)
Y0 = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, F, dev)


def run(cg_steps, short_max, iters=3):
    X, Y = X0.clone(), Y0.clone()
    eng = E.AlsEngine(Cui, Ciu, X, Y, 0.01, cg_steps, short_max=short_max)
    eng.iteration()
    torch.cuda.synchronize()
    eng.events = []
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.iteration()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    ms = {k: v[0] / v[1] for k, v in eng.event_times_ms().items()}
    return t, ms, X, Y


t_cg, ms_cg, _, _ = run(3, None)
print("CG-3   : %.2f ms per iteration  %s" % (t_cg * 1e3, {k: round(v, 2) for k, v in ms_cg.items()}), flush=True)
t_g, ms_g, _, _ = run(3, 0)
print("CG-3, every row through the explicit-Gram kernel: %.2f ms per iteration  %s" %
      (t_g * 1e3, {k: round(v, 2) for k, v in ms_g.items()}), flush=True)
t_ex, ms_ex, X, Y = run(-1, 0)
print("exact  : %.2f ms per iteration  %s" % (t_ex * 1e3, {k: round(v, 2) for k, v in ms_ex.items()}), flush=True)

# sampled user rows of one more exact user sweep against the float64-carried oracle
import oracle  # noqa: E402
import scipy.sparse as sp  # noqa: E402

rows = np.sort(np.random.default_rng(1).choice(U, 300, replace=False))
ptr = indptr.cpu().numpy()
idx = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows])
sub = sp.csr_matrix((Cui.data.cpu().numpy()[idx], Cui.indices.cpu().numpy()[idx],
                     np.concatenate([[0], np.cumsum(ptr[rows + 1] - ptr[rows])])), shape=(len(rows), I))
Yh = Y.cpu().numpy()[:, :f].copy()
ref = np.zeros((len(rows), f), dtype=np.float32)
oracle.als_half_sweep_exact(sub, ref, Yh, 0.01)
G = E.gram(Y, 0.01)
E.SweepPlan(Cui, F, 0, 0).sweep(X, Y, G, -1)
got = X.cpu().numpy()[rows][:, :f]
print("exact user sweep, 300 sampled rows: %.2e from the float64-carried solve" %
      (np.linalg.norm(got - ref) / np.linalg.norm(ref)), flush=True)
