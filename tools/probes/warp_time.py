"""Time one concurrent WARP epoch on the synthetic LightFM problem (users x items of the bench, sampled interactions)."""
import sys, os, time, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import LightFM, synth
U, I = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 27_000
p = synth.lightfm_problem(44, U, I)
inter = synth.interactions(42, U, I, 50.0).tocoo()
m = LightFM(no_components=15, loss="warp", random_state=42, learning_schedule="adagrad", item_alpha=1e-6)
t0 = time.time(); m.fit_partial(inter, user_features=p["user_features"], item_features=p["item_features"], epochs=1); t1 = time.time()
m.fit_partial(inter, user_features=p["user_features"], item_features=p["item_features"], epochs=1); t2 = time.time()
print(f"examples {inter.nnz}: first call {t1 - t0:.2f} s, second {t2 - t1:.2f} s (uploads and downloads included)")
