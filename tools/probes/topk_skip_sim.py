"""Could the all-users top-K sweep skip whole item tiles?  Replays the sweep in torch for a sample of users of the
benchmark's trained factors (ideal thresholds, liked items filtered) and counts the (wavefront, tile) pairs --
32 consecutive users, 32 consecutive positions -- in which

  * some score beats its user's threshold (the pairs that need the selection),
  * the Cauchy-Schwarz bound  |q| max|y|(tile) > threshold  holds for some user (the pairs that need any product),
  * the hi*hi plane product plus its error bound 2^-8 |q| max|y|(tile) beats the threshold for some user (the pairs
    that need the two products with the mid planes),

for the items by descending interactions and by descending norm.

usage: topk_skip_sim.py [iterations] [sample users]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
U, I, f, K = 1_000_000, 27_000, 128, 20
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
Cui = E.DeviceCSR(indptr, indices, data * 40.0, (U, I))
rs = np.random.RandomState(42)
X = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, f, dev)
Y = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, f, dev)
eng = E.AlsEngine(Cui, None, X, Y, 0.01, 3)
for _ in range(iters):
    eng.iteration()
torch.cuda.synchronize()
norms = Y.norm(dim=1)
pop = torch.bincount(indices.long(), minlength=I)
u0 = 500_000
Q = X[u0:u0 + S]
qn = Q.norm(dim=1)
scores = Q @ Y.T
coarse = Q.to(torch.bfloat16).float() @ Y.to(torch.bfloat16).float().T
liked = torch.zeros(S, I, dtype=torch.bool, device=dev)
lo, hi = int(indptr[u0]), int(indptr[u0 + S])
rows = torch.repeat_interleave(torch.arange(S, device=dev), indptr[u0 + 1:u0 + S + 1] - indptr[u0:u0 + S])
liked[rows, indices[lo:hi].long()] = True
scores = torch.where(liked, torch.full_like(scores, -float("inf")), scores)


def replay(order, tag):
    sc, co, nr = scores[:, order], coarse[:, order], norms[order]
    buf = torch.full((S, K), -float("inf"), device=dev)
    thr = torch.full((S,), -float("inf"), device=dev)
    ntile = (I + 31) // 32
    hot = cs = cr = 0
    for t in range(ntile):
        new = sc[:, 32 * t:32 * t + 32]
        tn = nr[32 * t:32 * t + 32].max()
        wave = lambda m: int(m.view(S // 32, 32).any(1).sum())
        hot += wave((new > thr[:, None]).any(1))
        cs += wave(qn * tn > thr)
        cr += wave(co[:, 32 * t:32 * t + 32].max(1).values + qn * tn * 2.0 ** -8 > thr)
        buf = torch.topk(torch.cat([buf, new], 1), K, dim=1).values
        thr = torch.maximum(thr, buf[:, -1])
    tot = ntile * (S // 32)
    print("%-28s pairs with a candidate %5.1f %%   not excluded by |q||y| %5.1f %%   not excluded by hi*hi %5.1f %%" % (
        tag, 100.0 * hot / tot, 100.0 * cs / tot, 100.0 * cr / tot), flush=True)


replay(torch.argsort(pop, descending=True, stable=True), "descending interactions")
replay(torch.argsort(norms, descending=True, stable=True), "descending norm")
replay(torch.arange(I, device=dev), "own order")
