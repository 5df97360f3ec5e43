"""How much selection work would an early threshold save the all-users top-K sweep?  Trains the benchmark's factors
for a few iterations, takes a sample of users and replays the sweep over the items in torch, with the ideal
threshold (the k-th best score met so far among the items the user has not liked), counting

  * the candidates per user: scores above the threshold when they are met;
  * the share of (wavefront, tile) pairs that hold a candidate at all (32 consecutive users, 32 consecutive items):
    only those enter the kernel's slow path;

for the items in their own order, in the order of descending norm, in the order of descending popularity, and in
their own order with the threshold started from the scores of the M items of largest norm: either the k-th best of
those the user has not liked, or -- without looking at the liked row -- the (k + L)-th best of all M for a user
with L liked items.

usage: topk_seed_sim.py [iterations] [sample users]
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
scores = X[u0:u0 + S] @ Y.T                                   # [S, I]
liked = torch.zeros(S, I, dtype=torch.bool, device=dev)
lo, hi = int(indptr[u0]), int(indptr[u0 + S])
rows = torch.repeat_interleave(torch.arange(S, device=dev), indptr[u0 + 1:u0 + S + 1] - indptr[u0:u0 + S])
liked[rows, indices[lo:hi].long()] = True
L = liked.sum(1)
scores = torch.where(liked, torch.full_like(scores, -float("inf")), scores)
print("users %d  liked per user: mean %.1f max %d" % (S, L.float().mean().item(), L.max().item()), flush=True)


def replay(order, thr0, tag):
    sc = scores[:, order]
    buf = torch.full((S, K), -float("inf"), device=dev)
    thr = thr0.clone()
    cand = torch.zeros(S, device=dev)
    hot = 0
    ntile = (I + 31) // 32
    early = 0
    for t in range(ntile):
        new = sc[:, 32 * t:32 * t + 32]
        c = new > thr[:, None]
        cand += c.sum(1)
        h = int(c.any(1).view(S // 32, 32).any(1).sum())
        hot += h
        if t < ntile // 8:
            early += h
        buf = torch.topk(torch.cat([buf, new], 1), K, dim=1).values
        thr = torch.maximum(thr, buf[:, -1])
    print("%-46s candidates/user %6.1f   hot (wave, tile) pairs %5.1f %%  (first eighth of the sweep: %5.1f %%)" % (
        tag, cand.mean().item(), 100.0 * hot / (ntile * (S // 32)), 100.0 * early / (ntile // 8 * (S // 32))), flush=True)


own = torch.arange(I, device=dev)
none = torch.full((S,), -float("inf"), device=dev)
replay(own, none, "own order")
replay(torch.argsort(norms, descending=True, stable=True), none, "descending norm")
replay(torch.argsort(pop, descending=True, stable=True), none, "descending popularity")
for M in (64, 128, 256, 512):
    top = torch.argsort(norms, descending=True, stable=True)[:M]
    seed = scores[:, top]                                     # liked ones are -inf here
    t_exact = torch.topk(seed, K, dim=1).values[:, -1]
    replay(own, t_exact, "own order, seed M=%d, liked rows read" % M)
    raw = X[u0:u0 + S] @ Y[top].T
    srt = torch.sort(raw, dim=1, descending=True).values
    r = K + L - 1
    t_blind = torch.where(r < M, srt.gather(1, r.clamp(max=M - 1)[:, None])[:, 0], none)
    replay(own, t_blind, "own order, seed M=%d, rank k+L" % M)
    topp = torch.argsort(pop, descending=True, stable=True)[:M]
    rawp = X[u0:u0 + S] @ Y[topp].T
    srtp = torch.sort(rawp, dim=1, descending=True).values
    t_blindp = torch.where(r < M, srtp.gather(1, r.clamp(max=M - 1)[:, None])[:, 0], none)
    replay(own, t_blindp, "own order, seed M=%d most popular, rank k+L" % M)
