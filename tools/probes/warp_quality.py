"""Does the concurrent WARP epoch learn as well as the serial one at scale?

usage: warp_quality.py [users items per_user clusters epochs]        (WARP_QUALITY_MODES=concurrent,serial)
Planted-cluster interactions over the hybrid model's feature layout (identity + 3 shared tokens per user, identity
+ tags per item); held-out AUC after each epoch for the concurrent mode and the serial mode (which equals the
sequential CPU statement bit for bit and takes about 32 us per example; default: only up to 30000 users).
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import LightFM, synth  # noqa: E402
from movie_recommender_system_amd.lightfm import auc_score  # noqa: E402

nu, ni, per_user, clusters, epochs = [int(v) for v in (sys.argv[1:6] + [20000, 2000, 25, 20, 3][len(sys.argv) - 1:])]
p = synth.lightfm_problem(9, nu, ni)
rng = np.random.default_rng(0)
cu, ci = rng.integers(0, clusters, nu), rng.integers(0, clusters, ni)
by_cluster = [np.where(ci == c)[0] for c in range(clusters)]
rows = np.repeat(np.arange(nu), per_user + 3)
cols = np.empty((nu, per_user + 3), dtype=np.int64)
for u in range(nu):
    cand = by_cluster[cu[u]]
    cols[u, :per_user] = cand[rng.integers(0, len(cand), per_user)]
    cols[u, per_user:] = rng.integers(0, ni, 3)
inter = sp.csr_matrix((np.ones(rows.size, np.float32), (rows, cols.ravel())), shape=(nu, ni))
inter.sum_duplicates()
inter.data[:] = 1.0
train, test = synth.split_holdout(inter, 3, 0.2)
train_coo, test = sp.coo_matrix(train), sp.csr_matrix(test)
print(f"{nu} users x {ni} items, {train_coo.nnz} train / {test.nnz} held-out interactions", flush=True)
modes = os.environ.get("WARP_QUALITY_MODES", "concurrent,serial" if nu <= 30000 else "concurrent").split(",")
for serial in [m == "serial" for m in modes]:
    m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=1e-6, serial=serial)
    for e in range(epochs):
        t0 = time.time()
        m.fit_partial(train_coo, user_features=p["user_features"], item_features=p["item_features"], epochs=1)
        dt = time.time() - t0
        auc = float(auc_score(m, test, sp.csr_matrix(train), user_features=p["user_features"],
                              item_features=p["item_features"]).mean())
        print(f"{'serial    ' if serial else 'concurrent'} epoch {e + 1}: held-out AUC {auc:.4f}  ({dt:.2f} s)", flush=True)
