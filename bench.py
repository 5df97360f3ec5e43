"""Benchmark of the recommender hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): implicit-feedback ALS with factors=128,
fp32, on synthetic 1 M users x 27 k movies (~5e7 interactions, seed 42), then
the top-20 recommend pass for all users from the trained factors.  One "step" is
one ALS iteration = user half-sweep + item half-sweep (both Grams and, with
N > 1, both all-gathers).  The problem size is fixed as N grows (strong
scaling): users and items are sharded across the ranks by stored entries.

Rank 0 prints ONE JSON line.  `value` is seconds per ALS iteration (max over
ranks, inputs resident in HBM).  `roofline` describes the dominant kernel (the
half-sweep kernel with the larger time per iteration): algorithmic bytes per
launch / live HIP-event time per launch against the 8 TB/s HBM peak.  `cpu_baseline` times the CPU oracle (a
restatement of implicit 0.7.2's CG solver) on a bounded sample of the same
matrix on the host cores.  `extra` carries the top-20 pass (users/s, fp32 MFMA
roofline) and the per-kernel times.
"""

from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=27_000)
    ap.add_argument("--mean-degree", type=float, default=50.0)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--alpha", type=float, default=40.0)
    ap.add_argument("--regularization", type=float, default=0.01)
    ap.add_argument("--topk", type=int, default=20)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--no-topk", action="store_true")
    ap.add_argument("--no-lightfm-train", action="store_true")
    ap.add_argument("--no-lightfm", action="store_true")
    ap.add_argument("--no-config1", action="store_true", help="skip the factors=64 figures beside the headline")
    ap.add_argument("--no-requests", action="store_true", help="skip the per-request latency figures")
    ap.add_argument("--no-matrix-build", action="store_true", help="skip the events -> CSR figures")
    return ap.parse_args()


def sweep_bytes(nnz: int, rows: int, f: int) -> float:
    """Algorithmic bytes of one half-sweep: every stored entry gathers one factor
    row (4f) plus its index and value (8); every solved row is read and written
    once (2*4f); the row pointers once."""
    return nnz * (4.0 * f + 8.0) + rows * 8.0 * f + (rows + 1) * 8.0


def _leading_rows(csr, rows):
    """Rows [0, rows) of a device CSR as a scipy matrix (only those entries cross PCIe)."""
    import scipy.sparse as sp
    ptr = (csr.indptr[: rows + 1] - csr.indptr[0]).cpu().numpy()
    e = int(ptr[-1])
    return sp.csr_matrix((csr.data[:e].cpu().numpy(), csr.indices[:e].cpu().numpy(), ptr), shape=(rows, csr.shape[1]))


def cpu_baseline(Cu, Ci, users, items, nnz, f, reg, budget_s, seed):
    """Time the oracle's half-sweeps on leading row blocks sized to the budget
    and extrapolate by stored entries to one full iteration.  `Cu` / `Ci`: device
    CSR of (leading) user rows and item rows; only the sampled rows are downloaded."""
    import oracle

    # timed, never compared here: the AVX-512 build of the oracle where the host has AVX-512
    march = oracle.use_widest_build()
    threads = oracle.num_threads()
    rs = np.random.RandomState(seed)
    # factors of the benchmark's start; only their size matters for the time
    X = rs.rand(min(users, 1 << 20), f).astype(np.float32) * 0.01
    if users > X.shape[0]:
        X = X[np.arange(users) % X.shape[0]]
    Y = rs.rand(items, f).astype(np.float32) * 0.01

    def timed(C, A, B, budget):
        n_rows_total = C.shape[0]
        G = (B.T @ B + reg * np.eye(f, dtype=np.float32)).astype(np.float32)
        probe = max(1, n_rows_total // 200)
        Cp = _leading_rows(C, probe)
        t0 = time.perf_counter()
        oracle.als_half_sweep(Cp, A, B, reg, 3, rows=(0, probe), YtY=G)
        tp = time.perf_counter() - t0
        rate = max(Cp.nnz, 1) / max(tp, 1e-6)           # entries / s
        ptr = (C.indptr - C.indptr[0]).cpu().numpy()
        rows = int(np.searchsorted(ptr, ptr[probe] + rate * budget))
        rows = max(min(rows, n_rows_total), min(probe + 1, n_rows_total))
        Cs = _leading_rows(C, rows)
        t0 = time.perf_counter()
        oracle.als_half_sweep(Cs, A, B, reg, 3, rows=(probe, rows), YtY=G)
        t = time.perf_counter() - t0
        return t, int(ptr[rows] - ptr[probe]), rows - probe

    tu, nnz_u, rows_u = timed(Cu, X, Y, budget_s / 2)
    ti, nnz_i, rows_i = timed(Ci, Y, X, budget_s / 2)
    est = tu * nnz / max(nnz_u, 1) + ti * nnz / max(nnz_i, 1)
    return {
        "value": est, "unit": "s/iteration", "cores": threads, "kind": "port",
        "sample": (f"oracle half-sweeps (restatement of implicit 0.7.2 CG, plain C loops built with gcc -O3 "
                   f"-march={march}, OpenMP {threads} threads) on "
                   f"{rows_u} user rows ({nnz_u} entries, {tu:.2f} s) and {rows_i} item rows ({nnz_i} entries, "
                   f"{ti:.2f} s) of the same matrix, f={f}; scaled by stored entries to one iteration"),
    }


def cpu_side_baselines(Cu, items, f, k, seed, n_users=4000):
    """The oracle's top-k pass (score chain + select, OpenMP) and its sequential WARP epoch on a bounded sample
    (a few seconds each), for the extra figures reported beside the device's."""
    import scipy.sparse as sp
    import oracle
    from movie_recommender_system_amd import synth

    rs = np.random.RandomState(seed)
    n_users = min(n_users, Cu.shape[0])
    Q = rs.rand(n_users, f).astype(np.float32)
    Y = rs.rand(items, f).astype(np.float32)
    liked = _leading_rows(Cu, n_users)
    liked.data[:] = 1.0
    t0 = time.perf_counter()
    oracle.topk(Q, Y, k, liked=liked)
    t_top = time.perf_counter() - t0
    prob = synth.lightfm_problem(44, n_users, items)
    ur = oracle.lightfm_repr(prob["user_features"], prob["user_embeddings"], prob["user_biases"])
    ir = oracle.lightfm_repr(prob["item_features"], prob["item_embeddings"], prob["item_biases"])
    t0 = time.perf_counter()
    oracle.lightfm_topk(ur, ir, k)
    t_lfm = time.perf_counter() - t0
    coo = liked.tocoo()
    st = oracle.lightfm_init(prob["user_features"].shape[1], prob["item_features"].shape[1], 15, seed)
    t0 = time.perf_counter()
    oracle.lightfm_warp_epoch(st, coo, prob["user_features"], prob["item_features"],
                              rs.permutation(coo.nnz).astype(np.int32), seed, item_alpha=1e-6)
    t_warp = time.perf_counter() - t0
    return {"cores": oracle.num_threads(), "sample_users": n_users,
            "top20_users_per_s": n_users / t_top, "lightfm_top20_users_per_s": n_users / t_lfm,
            "lightfm_warp_examples_per_s_one_thread": coo.nnz / t_warp}


def _percentiles(samples_s):
    a = np.sort(np.asarray(samples_s)) * 1e6
    return {"p50_us": float(a[len(a) // 2]), "p99_us": float(a[int(len(a) * 0.99)]), "mean_us": float(a.mean()),
            "calls": int(len(a))}


def request_latency(dev, X, Y, f, liked_dev, items, seed, calls=2000):
    """The reference's real serve path, one user per request (recommenders/als_recommender.py:181-186 and :237
    behind api.py:26-34; hybrid_docker/inference.py:156-165,197): wall time per call of the plugin's
    `recommend_for_user(N=20)` and `get_similar_items`, and of `compute_recommendations`, with everything resident
    on the device, against the same request answered on the host (the oracle's chain, and numpy's sgemv + partial
    sort, which is what implicit's CPU path amounts to).  The reference's published figure for this call is
    0.30 ms median / 0.37 ms mean at 11 557 items, f = 64 (notebooks/baseline_als.ipynb cell fd92678d)."""
    import scipy.sparse as sp
    import torch
    import oracle
    from movie_recommender_system_amd import ALSRecommender, AlternatingLeastSquares, LightFMScorer, artifacts, synth
    from movie_recommender_system_amd.lightfm import compute_recommendations

    out = {"reference_published": {"median_ms": 0.30, "mean_ms": 0.37, "items": 11557, "factors": 64,
                                   "source": "notebooks/baseline_als.ipynb cell fd92678d"}}
    rng = np.random.default_rng(seed)
    n_users = min(200_000, X.shape[0])           # users the plugin serves here (every request names one of them)
    M = _leading_rows(liked_dev, n_users)
    M.data[:] = 1.0
    pick = rng.integers(0, n_users, calls + 200)
    for ff in sorted({64, f}):
        model = AlternatingLeastSquares(factors=ff, device=dev)
        if ff == f:
            model.user_factors = X[:n_users, :f].contiguous().cpu().numpy()
            model.item_factors = Y[:, :f].contiguous().cpu().numpy()
        else:
            model.user_factors = rng.standard_normal((n_users, ff)).astype(np.float32) * 0.1
            model.item_factors = rng.standard_normal((items, ff)).astype(np.float32) * 0.1
        rec = ALSRecommender({"artefacts": {}, "metadata": {}, "device": str(dev)}, "als@bench")
        rec.model, rec.train_matrix = model, M
        rec.user_mappings = {"user_to_idx": {str(u): u for u in range(n_users)}, "n_users": n_users}
        rec.movie_mappings = {"idx_to_movie": {i: str(i) for i in range(items)},
                              "movie_to_idx": {str(i): i for i in range(items)}, "n_movies": items}
        rec.most_popular_movies = artifacts.popularity_order(M)
        rec.is_loaded = True
        rec.warmup()
        ts, tsim = [], []
        for j, u in enumerate(pick):
            t0 = time.perf_counter()
            res = rec.recommend_for_user(str(int(u)), N=20)
            ts.append(time.perf_counter() - t0)
            assert res.method == "ALS", rec.last_scoring_error
        for j, u in enumerate(pick[:600]):
            t0 = time.perf_counter()
            rec.get_similar_items(str(int(u) % items), N=20)
            tsim.append(time.perf_counter() - t0)
        # the same request on the host: the oracle (1 thread does all the work of a single query) and numpy
        # (two loops: the oracle's idle OpenMP threads spin for a while after a call and would sit on the
        # cores numpy's BLAS threads want)
        tor, tnp = [], []
        Yh, Xh = model.item_factors, model.user_factors
        for u in pick[:300]:
            row = M[int(u)]
            t0 = time.perf_counter()
            sc = Yh @ Xh[u]
            sc[row.indices] = -np.inf
            top = np.argpartition(-sc, 20)[:20]
            top = top[np.argsort(-sc[top])]
            tnp.append(time.perf_counter() - t0)
        for u in pick[:60]:
            row = M[int(u)]
            t0 = time.perf_counter()
            oid, _ = oracle.topk(Xh[u:u + 1], Yh, 20, liked=row)
            tor.append(time.perf_counter() - t0)
        got = rec.recommend_for_user(str(int(pick[0])), N=20)
        oid, _ = oracle.topk(Xh[pick[0]:pick[0] + 1], Yh, 20, liked=M[int(pick[0])])
        assert [r["movie_id"] for r in got.recommendations] == [str(int(i)) for i in oid[0]]
        out[f"als_f{ff}_{items}_items"] = {
            "recommend_for_user": _percentiles(ts[200:]), "get_similar_items": _percentiles(tsim[100:]),
            "cpu_oracle_one_thread": _percentiles(tor[5:]), "cpu_numpy_sgemv_argpartition": _percentiles(tnp[50:]),
            "scoring_failures": rec.scoring_failures}
        rec.teardown()
        del rec, model
    for ni in sorted({7208, items}):
        nu = 20_000
        prob = synth.lightfm_problem(44, nu, ni)
        sc = LightFMScorer(prob["user_embeddings"], prob["user_biases"], prob["item_embeddings"], prob["item_biases"],
                           device=dev)
        state = {"model": sc, "user_features": prob["user_features"], "item_features": prob["item_features"],
                 "inv_item_map": {i: i for i in range(ni)}, "user_to_idx": {u: u for u in range(nu)},
                 "user_feature_mapping": {}}
        ts = []
        for u in pick[:1200]:
            t0 = time.perf_counter()
            compute_recommendations(int(u) % nu, state, top_k=20)
            ts.append(time.perf_counter() - t0)
        ur = oracle.lightfm_repr(prob["user_features"][:64], prob["user_embeddings"], prob["user_biases"])
        ir = oracle.lightfm_repr(prob["item_features"], prob["item_embeddings"], prob["item_biases"])
        tor = []
        for u in range(40):
            t0 = time.perf_counter()
            oracle.lightfm_topk(ur[u:u + 1], ir, 20)
            tor.append(time.perf_counter() - t0)
        out[f"lightfm_nc15_{ni}_items"] = {"compute_recommendations": _percentiles(ts[200:]),
                                           "cpu_oracle_scoring_only_one_thread": _percentiles(tor[5:])}
        del sc, state, prob
    torch.cuda.synchronize()
    return out


def matrix_build(dev, csr, U, I, seed):
    """The step before the path (`build_matrix`, als/als.py:107-123: a pandas groupby over the watch events, then
    scipy's CSR constructor): the benchmark matrix's pairs, a tenth of them seen twice, in random order, turned
    into the binary CSR and the per-movie counts by `mirec_events_to_csr`; beside it the reference's own host
    statements on a sample of the same events, scaled by the event count."""
    import pandas as pd
    import scipy.sparse as sp
    import torch
    from movie_recommender_system_amd import engine as E

    rows = torch.repeat_interleave(torch.arange(U, dtype=torch.int32, device=dev), (csr.indptr[1:] - csr.indptr[:-1]))
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    again = torch.randint(0, rows.numel(), (rows.numel() // 10,), device=dev, generator=g)
    eu, ei = torch.cat([rows, rows[again]]), torch.cat([csr.indices, csr.indices[again]])
    perm = torch.randperm(eu.numel(), device=dev, generator=g)
    eu, ei = eu[perm].contiguous(), ei[perm].contiguous()
    del rows, again, perm
    n = int(eu.numel())
    E.DeviceCSR.from_events(eu[:1 << 20].contiguous(), ei[:1 << 20].contiguous(), (U, I), with_item_counts=True)
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out, counts = E.DeviceCSR.from_events(eu, ei, (U, I), with_item_counts=True)
        b.record()
        torch.cuda.synchronize()
        best = a.elapsed_time(b) if best is None else min(best, a.elapsed_time(b))
    assert out.nnz == csr.nnz
    # events in (8 B each), CSR out (index + value per pair, row pointers), counts: what any build must move
    moved = n * 8.0 + out.nnz * 8.0 + (U + 1) * 8.0 + I * 4.0
    ns = min(n, 5_000_000)
    hu, hi = eu[:ns].cpu().numpy(), ei[:ns].cpu().numpy()
    t0 = time.perf_counter()
    df = pd.DataFrame({"user_idx": hu, "movie_idx": hi})
    grouped = df.groupby(["user_idx", "movie_idx"]).size().reset_index(name="count")       # als/als.py:118
    m = sp.csr_matrix((np.ones(len(grouped), dtype=np.float32), (grouped["user_idx"].to_numpy(), grouped["movie_idx"].to_numpy())),
                      shape=(U, I))                                                        # als/als.py:119-122
    t_host = time.perf_counter() - t0
    del m, df, grouped
    return {"events": n, "pairs": int(out.nnz), "device_ms": best, "events_per_s": n / (best * 1e-3),
            "compulsory_GBps": moved / (best * 1e-3) / 1e9,
            "host_pandas_groupby_and_scipy": {"sample_events": ns, "seconds": t_host, "events_per_s": ns / t_host,
                                              "scaled_to_all_events_s": t_host * n / ns},
            "speedup_vs_host": (t_host * n / ns) / (best * 1e-3)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from movie_recommender_system_amd import engine as E
    from movie_recommender_system_amd import dist as D
    from movie_recommender_system_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # (the two overrides exist for rehearsing the multi-rank path on a one-GPU box: tools/rehearse_multi_rank.py)
    local_rank = int(os.environ.get("MIREC_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MIREC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    U, I, f = args.users, args.items, args.factors
    F = E.factor_width(f)
    t_setup = time.perf_counter()
    # Storage is sharded like the solves: a rank generates the rows of its own user blocks only; the rows of the
    # transposed matrix for its item range reach it through one all-to-all inside gpu_driver.
    layout = D.user_layout(U, world)
    shards = []
    for a, b in layout.owned(rank):
        ip, ix, dt = synth.interactions_torch(args.seed, U, I, args.mean_degree, dev, users=(a, b))
        dt *= args.alpha                       # fit() multiplies the stored values by alpha
        shards.append((a, b, E.DeviceCSR(ip, ix, dt, (b - a, I))))
    nnz_local = sum(c.nnz for _, _, c in shards)
    nnz = nnz_local
    if world > 1:
        t = torch.tensor([nnz_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        nnz = int(t.item())
    rs = np.random.RandomState(args.seed)

    def init_factors(n):
        """`rs.rand(n, f).astype(f32) * 0.01` (implicit's start), drawn in slabs so that the float64 draws of a
        10 M-row matrix never sit in host memory at once (the stream, and so the values, are the same)."""
        out = torch.zeros((n, F), dtype=torch.float32, device=dev)
        step = 1 << 20
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            out[r0:r1, :f] = torch.from_numpy(rs.rand(r1 - r0, f).astype(np.float32) * 0.01).to(dev)
        return out

    X = init_factors(U)
    Y = init_factors(I)
    drv, eng = D.gpu_driver(shards, None, X, Y, args.regularization, 3, rank, world)
    X = drv.X  # the driver's padded copy is the user factor matrix from here on
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    for _ in range(args.warmup):
        drv.iteration()
    eng.events = []
    if world > 1:
        drv.events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        drv.iteration()
    barrier()
    elapsed = time.perf_counter() - t0
    times = eng.event_times_ms()
    eng.events = None
    exchange = {k: v[0] / max(v[1], 1) for k, v in drv.exchange_times_ms().items()}
    drv.events = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    s_per_iter = elapsed / args.steps

    # ---- roofline: one entry per half-sweep kernel (live HIP-event time, algorithmic bytes of the rows that
    # kernel solved); the dominant one (more time per iteration) is reported as `roofline`.  On one GPU a
    # half-sweep is one launch of its kernel; on N GPUs the user side is cut into blocks (one launch each, the
    # exchange of a block overlaps the next), and the figures are then per half-sweep: bytes of all of this
    # rank's blocks over the summed duration of their launches.
    u_rows, u_nnz, u_short, u_chunks = eng.user_totals()

    def split_by_kernel(plans):
        """(rows, entries) solved by the matrix-free kernel and by the explicit-Gram kernel over these plans."""
        rs = ns = rm = nm = 0
        for p in plans:
            if p is None:
                continue
            ln = p.csr.indptr[1:] - p.csr.indptr[:-1]
            short = ln <= p.short_max
            rs += int(short.sum())
            ns += int(ln[short].sum())
            rm += int((~short).sum())
            nm += int(ln[~short].sum())
        return (rs, ns), (rm, nm)

    def side(tag, plans):
        out = []
        t_s, c_s = times.get(tag + "_short", (0.0, 0))
        t_m, c_m = times.get(tag + "_main", (0.0, 0))
        (rs, ns), (rm, nm) = split_by_kernel(plans)
        # algorithmic bytes of the rows each kernel solved
        if c_s:
            # widths 64 and 128 run the variant whose G v products are evaluated on the matrix cores
            out.append(("als_short_mfma_kernel" if F in (64, 128) else "als_short_kernel", sweep_bytes(ns, rs, F),
                        t_s / args.steps, c_s))
        if c_m:
            out.append(("als_gram_cg_kernel", sweep_bytes(nm, rm, F), t_m / args.steps, c_m))
        return out

    # HBM-side bytes per launch cannot be read from inside this process: they come from separate
    # rocprofv3 --pmc passes over the same workload, whose summary is committed next to the kernel
    # statistics.  They are attached only when workload and world size match that capture.
    def pmc_traffic(tag, kernel):
        if world != 1:
            return None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
            try:
                d = json.load(open(path))
            except (OSError, ValueError):
                continue
            w = d.get("workload", {})
            if (w.get("users"), w.get("items"), w.get("factors"), w.get("seed"), w.get("mean_degree")) != \
                    (U, I, f, args.seed, args.mean_degree):
                continue
            e = d.get(tag, {}).get("mirec::" + kernel, {})
            if "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"]
            break  # the newest capture of this workload has no such kernel: nothing older applies
        return None

    entries = [(tag, k, b, ms, n) for tag, plans in (("user", eng.plans_u), ("item", [eng.plan_i]))
               for (k, b, ms, n) in side(tag, plans)]
    rl_all = []
    for tag, k, b, ms, n in entries:
        gbs = (b / (ms * 1e-3) / 1e9) if (b and ms > 0) else None
        # second template argument: 64-bit addressing of the gathered table (4 GiB and more); the explicit-Gram
        # kernel has a third: the instantiation with the exact solve in place of the CG steps (not the one timed here)
        wide = "true" if (I if tag == "user" else U) * F * 4 > (1 << 32) else "false"
        kname = f"{k}<{F // 32}, {wide}>" if k != "als_gram_cg_kernel" else f"{k}<{F // 32}, {wide}, false>"
        rl_all.append({"side": tag, "kernel": kname, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": (gbs / HBM_PEAK_GBS) if gbs else None,
                       "traffic": pmc_traffic(tag, kname),
                       "bytes_per_launch": b, "avg_launch_ms": ms, "launches_timed": n})
        # (per half-sweep of this rank; equal to per launch when the side is one block, as on one GPU)
    dom = max(rl_all, key=lambda e: e["avg_launch_ms"]) if rl_all else None
    roofline = dict(dom) if dom else None
    kernel_ms = {k: v[0] / max(v[1], 1) for k, v in times.items()}

    extra = {"setup_s": t_setup, "nnz": nnz, "kernel_ms_per_launch": kernel_ms, "half_sweep_rooflines": rl_all,
             "iteration_GBps": (sweep_bytes(u_nnz, u_rows, F) +
                                sweep_bytes(eng.Ciu_local.nnz, eng.Ciu_local.shape[0], F)) / s_per_iter / 1e9,
             "user_short_rows": u_short, "item_short_rows": eng.plan_i.n_short_rows,
             "user_chunks": u_chunks, "item_chunks": eng.plan_i.n_chunks,
             "user_blocks": len(eng.user_blocks),
             # N > 1: what this rank's compute stream spends per iteration waiting for the gathers of the user slabs
             # that its solves did not hide, and on the item exchange
             "exchange_ms_per_iteration": exchange,
             "item_split_rows": eng.plan_i.n_split_rows,
             # interaction-matrix storage on this rank (both orientations) against the whole matrix's
             "resident_matrix_bytes": eng.resident_matrix_bytes(),
             "whole_matrix_bytes": 2 * (nnz * 8) + (U + I + 2) * 8}

    # ---- top-K pass for this rank's users (no exchange needed): one launch per owned block of users
    if not args.no_topk:
        W = E.factor_width(f, for_scoring=True)
        Ys = Y if W == F else Y[:, :W].contiguous()
        jobs = []
        for a, b, c in shards:       # liked items = this rank's own rows of the interaction matrix
            if b > a:
                Q = X[a:b] if W == F else X[a:b, :W].contiguous()
                jobs.append((Q, c))
        n_local = sum(q.shape[0] for q, _ in jobs)
        # the sweep order of the pass belongs to the interaction matrix, not to the factors: built once per
        # matrix (timed on its own), outside the pass that is repeated per model
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _, m in jobs:
            m.build_order()
        torch.cuda.synchronize()
        extra["top20_order_build_ms"] = (time.perf_counter() - t0) * 1e3
        # one untimed pass first, as the iterations have theirs: code objects loaded, and the output and workspace
        # blocks (220 MB at the default size) in torch's caching allocator instead of a fresh device allocation inside
        # the first timed pass
        outs = [E.topk(Q, Ys, args.topk, m) for Q, m in jobs]
        del outs
        barrier()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            outs = [E.topk(Q, Ys, args.topk, m) for Q, m in jobs]
        barrier()
        tk = (time.perf_counter() - t0) / reps
        if world > 1:
            t = torch.tensor([tk], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tk = float(t.item())
        flops = 2.0 * n_local * I * W
        extra["top20_users_per_s"] = U / tk
        extra["top20_pass_ms"] = tk * 1e3
        # widths that are multiples of 16 up to 128 with k <= 24 sweep on bf16 matrix-core scores -- the product of the
        # hi planes for every tile, the two products with a mid plane only for the tiles that can still hold a
        # candidate -- and re-score the survivors in fp32; everything else runs on the fp32 matrix pipe
        planes = W % 16 == 0 and W <= 128 and args.topk <= 24
        peak = MFMA_BF16_PEAK_TFLOPS if planes else MFMA_F32_PEAK_TFLOPS
        kname = ("topk_planes_kernel<%d, 8, 48, 6, true>" % W) if planes else "topk_kernel<%d, %s>" % (
            W, "4, 56" if args.topk <= 24 else "4, 128" if args.topk <= 96 else "2, 256")
        extra["topk_roofline"] = {"kernel": kname, "bound": "mfma", "achieved": flops / tk / 1e12, "peak": peak,
                                  "unit": "TFLOP/s", "frac": flops / tk / 1e12 / peak,
                                  "traffic": pmc_traffic("topk", kname) if args.topk == 20 else None,
                                  "note": "achieved = 2*users*items*f / time; peak = dense bf16 peak: one plane product "
                                          "per multiply-add of the score matrix is the least the sweep can issue (the "
                                          "figure of earlier rounds was against a third of it, for three products)"
                                  if planes else "achieved = 2*users*items*f / time; peak = dense fp32 matrix peak"}
        if planes:
            extra["topk_roofline"]["frac_of_three_product_peak"] = flops / tk / 1e12 / (peak / 3.0)
        del outs

    # ---- LightFM hybrid scoring (configs[4]): segment-sum + dot + top-20 for all users, rank 0's share
    if not args.no_lightfm and rank == 0:
        from movie_recommender_system_amd import LightFMScorer
        nu_l = U // world
        prob = synth.lightfm_problem(44, nu_l, I)
        sc = LightFMScorer(prob["user_embeddings"], prob["user_biases"], prob["item_embeddings"], prob["item_biases"],
                           device=dev)
        sc.recommend(None, prob["user_features"], prob["item_features"], args.topk)   # warm: pinned staging, code objects
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.recommend(None, prob["user_features"], prob["item_features"], args.topk)
        torch.cuda.synchronize()
        tl = time.perf_counter() - t0
        extra["lightfm_top20_users_per_s_per_gpu_incl_transfers"] = nu_l / tl
        # the two kernels of the pass on resident inputs, timed with HIP events on their stream: the feature
        # segment-sum (HBM roof) and the fused score + top-k at the padded scoring width (matrix roof)
        nc = sc.no_components
        Wl = E.scoring_width(nc + 2, args.topk)
        tabs = sc._tables(dev)
        ufc = E.DeviceCSR.from_scipy(prob["user_features"], dev)
        ifc = E.DeviceCSR.from_scipy(prob["item_features"], dev)
        from movie_recommender_system_amd import _lib as _L
        lib = _L.require()

        def seg(csr, emb, bias, ld, ec, bc, oc):
            out = torch.zeros((csr.shape[0], ld), dtype=torch.float32, device=dev)
            _L.check(lib.mirec_lightfm_repr(E._ptr(csr.indptr), E._ptr(csr.indices), E._ptr(csr.data), csr.shape[0],
                                            E._ptr(emb), E._ptr(bias), 1.0, nc, E._ptr(out), ld, ec, bc, oc,
                                            E._stream_ptr()), "mirec_lightfm_repr")
            return out

        def timed(fn, reps=3):
            fn()
            best = None
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                r = fn()
                b.record()
                torch.cuda.synchronize()
                best = a.elapsed_time(b) if best is None else min(best, a.elapsed_time(b))
            return best, r

        t_seg, Ql = timed(lambda: seg(ufc, tabs["ue"], tabs["ub"], Wl, 2, 0, 1))
        _, Yl = timed(lambda: seg(ifc, tabs["ie"], tabs["ib"], Wl, 2, 1, 0), reps=1)
        t_top, _ = timed(lambda: E.topk(Ql, Yl, args.topk, path="sweep"))
        seg_bytes = ufc.nnz * ((nc + 1) * 4 + 8) + nu_l * (Wl * 4 + 8)      # gathered table rows + index/weight + output rows
        useful = 2.0 * nu_l * I * (nc + 2)
        extra["lightfm_roofline"] = {
            "users": nu_l, "items": I, "no_components": nc, "scoring_width": Wl,
            "topk_kernel": "topk_planes_kernel<%d, 8, 48, 6, true>" % Wl, "topk_kernel_ms": t_top,
            "users_per_s_kernel_only": nu_l / ((t_top + t_seg) * 1e-3),
            "users_per_s_incl_transfers": nu_l / tl,
            "bound": "mfma", "achieved": useful / (t_top * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": useful / (t_top * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "note": "achieved = 2*users*items*(nc+2) useful flops / kernel time (the operand rows are padded to "
                    "scoring_width columns); peak = dense bf16 peak, one plane product per multiply-add (earlier "
                    "rounds: a third of it); the sweep at this width is bound by the instructions beside its two "
                    "matrix instructions per tile, not by the matrix pipe",
            "segment_sum": {"kernel": "lightfm_repr_kernel", "ms": t_seg, "bound": "hbm", "bytes": seg_bytes,
                            "achieved": seg_bytes / (t_seg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": seg_bytes / (t_seg * 1e-3) / 1e9 / HBM_PEAK_GBS}}
        del sc, Ql, Yl, ufc, ifc
        # WARP training, one epoch over the interactions of the first 100 k users (the second of two calls is timed;
        # feature / example uploads and the parameter download are inside the figure)
        if not args.no_lightfm_train:
            import scipy.sparse as sp
            from movie_recommender_system_amd import LightFM
            nu_t = min(100_000, nu_l, shards[0][2].shape[0])
            lead = _leading_rows(shards[0][2], nu_t)
            lead.data[:] = 1.0
            coo = lead.tocoo()
            uf_t = prob["user_features"][:nu_t]
            m = LightFM(no_components=15, loss="warp", random_state=args.seed, learning_schedule="adagrad",
                        item_alpha=1e-6, device=dev)
            t0 = time.perf_counter()
            m.fit_partial(coo, user_features=uf_t, item_features=prob["item_features"], epochs=1)
            # (a fresh model ramps its concurrency up from one workgroup; this first call also pays one-off set-up)
            extra["lightfm_warp_first_epoch_examples_per_s_incl_transfers"] = coo.nnz / (time.perf_counter() - t0)
            t0 = time.perf_counter()
            m.fit_partial(coo, user_features=uf_t, item_features=prob["item_features"], epochs=1)
            extra["lightfm_warp_examples_per_s_incl_transfers"] = coo.nnz / (time.perf_counter() - t0)
            extra["lightfm_warp_examples"] = int(coo.nnz)
            del m, coo
        del prob

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        Cu0 = shards[0][2]
        cpu = cpu_baseline(Cu0, eng.Ciu_local, U, I, nnz, f, args.regularization, args.cpu_seconds, args.seed)
        extra["speedup_vs_cpu_baseline"] = cpu["value"] / s_per_iter
        extra["cpu_oracle_other_passes"] = cpu_side_baselines(Cu0, I, f, args.topk, args.seed)
        if not args.no_requests:
            extra["request_latency"] = request_latency(dev, X, Y, f, Cu0, I, args.seed)
        if not args.no_matrix_build and nnz < 400_000_000:
            extra["matrix_build"] = matrix_build(dev, Cu0, U, I, args.seed)
        if f != 64 and not args.no_config1:
            # BASELINE configs[0]: the same matrix at factors = 64 -- the CPU oracle on a bounded sample and the
            # device iteration beside it (same shards and plans' policy, factor width 64)
            F1 = E.factor_width(64)
            rs1 = np.random.RandomState(args.seed)
            X1 = E.pad_factors(rs1.rand(U, 64).astype(np.float32) * 0.01, F1, dev)
            Y1 = E.pad_factors(rs1.rand(I, 64).astype(np.float32) * 0.01, F1, dev)
            eng1 = E.AlsEngine(None, None, X1, Y1, args.regularization, 3, user_blocks=[(a, b) for a, b, _ in shards],
                               user_shards=[c for _, _, c in shards], item_shard=eng.Ciu_local)
            for _ in range(2):
                eng1.iteration()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n1 = 5
            for _ in range(n1):
                eng1.iteration()
            torch.cuda.synchronize()
            t64 = (time.perf_counter() - t0) / n1
            del eng1, X1, Y1
            cpu64 = cpu_baseline(Cu0, eng.Ciu_local, U, I, nnz, 64, args.regularization,
                                 min(args.cpu_seconds, 8.0), args.seed)
            extra["config1_factors64"] = {
                "gpu_s_per_iteration": t64, "cpu_baseline": cpu64, "speedup_vs_cpu_baseline": cpu64["value"] / t64,
                "algorithmic_GBps": (sweep_bytes(nnz, U, F1) + sweep_bytes(nnz, I, F1)) / t64 / 1e9}

    if rank == 0:
        out = {
            "metric": "als_seconds_per_iteration", "value": s_per_iter, "unit": "s/iteration", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": s_per_iter * 1e3, "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # factors, sums and solves are fp32; where a product runs on the matrix cores (the Gram of a long row, G v of
            # a short one) each fp32 operand enters as two fp16 planes, 22 significant bits, accumulated in fp32
            "operand_bits": {"vector_alu": 24, "matrix_core_products": 22, "accumulate": 24},
            "config": {"workload": f"implicit ALS (CG-3) f={f} fp32 (matrix-core operands as two fp16 planes = 22 bits), {U} users x {I} movies, {nnz} interactions, "
                                   f"alpha={args.alpha:g}, reg={args.regularization:g}; then top-{args.topk} for all users",
                       "users": U, "items": I, "nnz": nnz, "factors": f, "sharding": f"users block-cyclic, items by stored entries, x{world}; interaction rows stored per rank"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 did the single-rank extras above: all ranks tear the group down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
