"""One row of one randomised half-sweep case, step by step: distance of the device's iterate from float64 CG after 0..3
steps, by component band, with the whole matrix and with the row alone.

usage: debug_sweep_row.py <libmirec.so> <seed> <case> <row>
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from movie_recommender_system_amd import _lib  # noqa: E402

_lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E  # noqa: E402
import als_numpy  # noqa: E402

seed, case, row = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng([seed, case])
f = int(rng.choice([32, 64, 96, 128, 160, 192]))
rows = int(rng.choice([1, 2, 15, 16, 17, 257, int(rng.integers(1, 600))]))
cols = int(rng.choice([1, 3, 40, 190, int(rng.integers(1, 900))]))
dens = float(rng.choice([0.0, 0.01, 0.06, 0.3, 1.0]))
chunk = int(rng.choice([0, 32, 64, 200]))
short_max = int(rng.choice([0, 1, 8, 40, 1 << 30]))
neg = bool(rng.random() < 0.5)
m = sp.random(rows, cols, density=dens, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr",
              dtype=np.float32)
m.data[:] = rng.uniform(0.5, 40.0, size=m.nnz).astype(np.float32)
if neg:
    m.data[rng.random(m.nnz) < 0.2] *= -1
m.sort_indices()
reg = float(rng.choice([0.01, 0.1, 1.0]))
X = (rng.random((rows, f)) * 0.1).astype(np.float32)
Y = (rng.standard_normal((cols, f)) * 0.1).astype(np.float32)
print(f"rows={rows} cols={cols} f={f} dens={dens} chunk={chunk} short_max={short_max} neg={neg} reg={reg}")
ln = np.diff(m.indptr)
print("row", row, "entries", ln[row], "indices", m.indices[m.indptr[row]:m.indptr[row + 1]].tolist(), "values",
      m.data[m.indptr[row]:m.indptr[row + 1]].tolist())
order = np.argsort(-ln, kind="stable")
print("position of the row by descending length:", int(np.nonzero(order == row)[0][0]), "of", rows)
dev = torch.device("cuda", 0)
dY = torch.from_numpy(Y).to(dev)
G = E.gram(dY, reg)


def run(mat, steps):
    dC = E.DeviceCSR.from_scipy(mat, dev)
    dX = torch.from_numpy(X).to(dev)
    E.SweepPlan(dC, f, chunk, short_max).sweep(dX, dY, G, steps)
    return dX.cpu().numpy()


alone = m.copy().tolil()
for r in range(rows):
    if r != row:
        alone.rows[r] = []
        alone.data[r] = []
alone = alone.tocsr().astype(np.float32)
alone.sort_indices()
for steps in (0, 1, 2, 3):
    X64 = als_numpy.half_sweep_f64(m, X, Y, reg, steps)
    for name, mat in (("whole", m), ("alone", alone)):
        out = run(mat, steps)
        d = out[row] - X64[row]
        bands = np.abs(d).reshape(-1, 16).max(axis=1)
        print(f"steps {steps} {name}: |x64| {np.linalg.norm(X64[row]):.3e}  |device - x64| {np.linalg.norm(d):.3e}  per band of 16: "
              + " ".join(f"{b:.1e}" for b in bands))

# One step: x1 - x0 = alpha * r0.  Which part of r0 = b - (G + S) x0 is off?  Least squares of the device's step on
# the float64 r0, G x0 and the gathered rows.
Y64 = Y.astype(np.float64)
G64 = Y64.T @ Y64 + reg * np.eye(f)
A, b = als_numpy.row_system(m, row, Y64, G64)
x0 = X[row].astype(np.float64)
r0 = b - A @ x0
alpha = (r0 @ r0) / (r0 @ A @ r0)
d1 = run(alone, 1)[row].astype(np.float64) - x0
idx = m.indices[m.indptr[row]:m.indptr[row + 1]]
basis = np.stack([r0, G64 @ x0] + [Y64[k] for k in idx], axis=1)
coef, res, *_ = np.linalg.lstsq(basis, d1, rcond=None)
print("float64 alpha", alpha, " |r0|", np.linalg.norm(r0), " |G x0|", np.linalg.norm(G64 @ x0))
print("device step = %.6g * r0 + %.4g * G x0 + gathered rows * %s; unexplained %.3e of %.3e" % (
    coef[0], coef[1], np.array2string(coef[2:], precision=3), np.linalg.norm(d1 - basis @ coef), np.linalg.norm(d1)))
print("cosine(device step, r0) - 1 = %.3e" % (d1 @ r0 / np.linalg.norm(d1) / np.linalg.norm(r0) - 1.0))
a_dev = (d1 @ r0) / (r0 @ r0)
pAp, pAp_dev = r0 @ A @ r0, (r0 @ r0) / a_dev
print("alpha: device %.8g float64 %.8g;  p'Ap: device %.8g float64 %.8g  difference %.6g" % (a_dev, alpha, pAp_dev, pAp, pAp_dev - pAp))
print("candidates: reg |p|^2 %.6g   p'Gp %.6g   per entry (|c|-1)(y.p)^2: %s" % (
    reg * (r0 @ r0), r0 @ G64 @ r0,
    np.array2string(np.array([(abs(c) - 1) * (Y64[k] @ r0) ** 2 for k, c in zip(idx, m.data[m.indptr[row]:m.indptr[row + 1]])]), precision=6)))
