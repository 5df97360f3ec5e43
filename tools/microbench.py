"""Kernel-level timing helper for the GPU box (not part of the product or of bench.py's contract).

    python tools/microbench.py topk  [--users N --items N --factors F --k K --no-mask]
    python tools/microbench.py sweep [--users N --items N --factors F --side user|item --short-max S --chunk C]
"""

import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from movie_recommender_system_amd import engine as E  # noqa: E402
from movie_recommender_system_amd import synth  # noqa: E402


def timeit(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["topk", "sweep"])
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=27_000)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--mean-degree", type=float, default=50.0)
    ap.add_argument("--no-mask", action="store_true")
    ap.add_argument("--side", default="item")
    ap.add_argument("--short-max", type=int, default=-1)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--trained", type=int, default=0, help="topk: factors from this many ALS iterations, not random")
    ap.add_argument("--order", action="store_true", help="topk: sweep the items by descending interactions")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    U, I, f = a.users, a.items, a.factors
    indptr, indices, data = synth.interactions_torch(42, U, I, a.mean_degree, dev)
    Cui = E.DeviceCSR(indptr, indices, data * 40.0, (U, I))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    if a.what == "topk":
        W = E.factor_width(f, True)
        X = torch.randn((U, W), device=dev, generator=g) * 0.3
        Y = torch.randn((I, W), device=dev, generator=g) * 0.3
        if a.trained:
            import numpy as np
            rs = np.random.RandomState(42)
            X = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, W, dev)
            Y = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, W, dev)
            eng = E.AlsEngine(Cui, None, X, Y, 0.01, 3)
            for _ in range(a.trained):
                eng.iteration()
            torch.cuda.synchronize()
        mask = None if a.no_mask else E.DeviceCSR(indptr, indices, data, (U, I))
        if a.order and mask is not None:
            mask.build_order()
        mn, av = timeit(lambda: E.topk(X, Y, a.k, mask), a.reps)
        fl = 2.0 * U * I * W
        print(f"topk U={U} I={I} f={W} k={a.k} mask={not a.no_mask}: min {mn:.2f} ms avg {av:.2f} ms  "
              f"{U / mn * 1e3 / 1e6:.2f} M users/s  {fl / mn / 1e9:.1f} TFLOP/s")
    else:
        F = E.factor_width(f)
        X = torch.rand((U, F), device=dev, generator=g) * 0.1
        Y = torch.rand((I, F), device=dev, generator=g) * 0.1
        if a.side == "user":
            csr, rows, other = Cui, X, Y
        else:
            csr, rows, other = Cui.transpose(), Y, X
        G = E.gram(other, 0.01)
        plan = E.SweepPlan(csr, F, a.chunk, None if a.short_max < 0 else a.short_max)
        mn, av = timeit(lambda: plan.sweep(rows, other, G, 3), a.reps)
        by = csr.nnz * (4.0 * F + 8) + csr.shape[0] * 8.0 * F
        print(f"sweep side={a.side} rows={csr.shape[0]} nnz={csr.nnz} F={F} short_max={plan.short_max} "
              f"chunks={plan.n_chunks} split={plan.n_split_rows} short_rows={plan.n_short_rows}: "
              f"min {mn:.2f} ms avg {av:.2f} ms  {by / mn / 1e6:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()
