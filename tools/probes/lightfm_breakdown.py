"""Where the time of LightFMScorer.recommend(all users) goes at 1 M users x 27 k items."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import LightFMScorer, engine as E, synth

dev = torch.device("cuda", 0)
U, I = 1_000_000, 27_000
p = synth.lightfm_problem(44, U, I)
sc = LightFMScorer(p["user_embeddings"], p["user_biases"], p["item_embeddings"], p["item_biases"], device=dev)
sc.recommend(list(range(2048)), p["user_features"], p["item_features"], 20)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ids, scores = sc.recommend(None, p["user_features"], p["item_features"], 20)
    torch.cuda.synchronize()
    print(f"recommend(all): {(time.perf_counter() - t0) * 1e3:.1f} ms")
# pieces
t0 = time.perf_counter(); uf = sc._check_features(p["user_features"], U, p["user_embeddings"].shape[0], "user"); t1 = time.perf_counter()
print(f"check_features(user): {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); d = E.DeviceCSR.from_scipy(uf, dev); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"upload user features: {(t1 - t0) * 1e3:.1f} ms  ({(uf.indptr.nbytes * 2 + uf.indices.nbytes + uf.data.nbytes) / 1e6:.0f} MB)")
out = torch.empty((U, 20), dtype=torch.int32, device=dev)
t0 = time.perf_counter(); h = out.cpu(); t1 = time.perf_counter()
print(f"download 80 MB pageable: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); pin = torch.empty((U, 20), dtype=torch.int32).pin_memory(); t1 = time.perf_counter()
print(f"pin 80 MB: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); pin.copy_(out, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"download 80 MB pinned: {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); a = pin.numpy().copy(); t1 = time.perf_counter()
print(f"host copy 80 MB: {(t1 - t0) * 1e3:.1f} ms")
# the same through the ALS model's recommend_all
from movie_recommender_system_amd import AlternatingLeastSquares
import scipy.sparse as spx
rng = np.random.default_rng(0)
m = AlternatingLeastSquares(factors=128, device=dev)
m.user_factors = (rng.standard_normal((U, 128)) * 0.3).astype(np.float32)
m.item_factors = (rng.standard_normal((I, 128)) * 0.3).astype(np.float32)
ip, ix, dt = synth.interactions_torch(42, U, I, 50.0, dev)
M = spx.csr_matrix((dt.cpu().numpy(), ix.cpu().numpy(), ip.cpu().numpy()), shape=(U, I))
for rep in range(3):
    t0 = time.perf_counter()
    ids, scores = m.recommend_all(M, N=20)
    print(f"ALS recommend_all incl. uploads and downloads: {(time.perf_counter() - t0) * 1e3:.1f} ms")
chk = rng.integers(0, U, 50)
import oracle
oid, osc = oracle.topk(m.user_factors[chk], m.item_factors, 20, liked=M[chk])
print("sampled rows equal the oracle:", np.array_equal(ids[chk], oid) and np.array_equal(scores[chk], osc))
