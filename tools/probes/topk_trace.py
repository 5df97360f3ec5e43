"""Per-tile trace of one workgroup of the bf16-plane top-K kernel (library built with -DMIREC_DEBUG_TIMING)."""
import ctypes, sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import _lib
_lib.load(os.environ.get("MIREC_LIB", _lib.LIB_PATH))
from movie_recommender_system_amd import engine as E, synth
dev = torch.device("cuda", 0)
U, I, F = 1_000_000, 27_000, 128
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
Cui = E.DeviceCSR(indptr, indices, data, (U, I))
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((U, F), device=dev, generator=g) * 0.3
Y = torch.randn((I, F), device=dev, generator=g) * 0.3
lib = _lib.require()
buf = (ctypes.c_uint * (20 * 1024))()
for wg in [int(a) for a in sys.argv[1:]] or [0]:
    lib.mirec_topk_debug_mode(wg << 8)
    E.topk(X, Y, 20, Cui if os.environ.get("TRACE_MASK") else None); torch.cuda.synchronize()
    lib.mirec_topk_debug_trace(buf)
    tr = np.frombuffer(buf, dtype=np.uint32).reshape(20, 1024)[:, :844].astype(np.int64)
    sel, tile, stage, mat = tr[0:8], tr[8], tr[9:13], tr[16:20]
    print(f"workgroup {wg}: tile total mean {tile.mean():.0f}; matrix mean {mat.mean():.0f}; staging mean {stage.mean():.0f}; "
          f"selection mean {sel.mean():.0f}, mean of per-tile max {sel.max(0).mean():.0f}")
    for lo, hi in [(0, 8), (8, 32), (32, 100), (100, 300), (300, 600), (600, 844)]:
        sl = slice(lo, hi)
        print(f"  tiles {lo:3d}-{hi:3d}: tile {tile[sl].mean():7.0f}  matrix {mat[:, sl].mean():6.0f}  staging {stage[:, sl].mean():6.0f}  "
              f"selection mean {sel[:, sl].mean():6.0f} max-of-8 {sel[:, sl].max(0).mean():6.0f}  p90 of max {np.percentile(sel[:, sl].max(0), 90):6.0f}")
    a, b, cc = tr[13], tr[14], tr[15]
    for lo, hi in [(1, 8), (8, 32), (32, 100), (100, 300), (300, 600), (600, 844)]:
        sl = slice(lo, hi)
        print(f"  selector 0, tiles {lo:3d}-{hi:3d}: walk + score read {a[sl].mean():6.0f}  row loop {b[sl].mean():6.0f}  compactions + deferred {cc[sl].mean():6.0f} (nonzero-ish in {np.mean(cc[sl] > 400):.2f} of tiles)")
    np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", f"topk_trace_{wg}.npy"), tr)
lib.mirec_topk_debug_mode(0)
