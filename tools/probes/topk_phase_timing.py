"""Phase timing of the bf16-plane top-K kernel (library built with -DMIREC_DEBUG_TIMING)."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth
dev = torch.device("cuda", 0)
U, I, F = 1_000_000, 27_000, 128
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
Cui = E.DeviceCSR(indptr, indices, data, (U, I))
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn((U, F), device=dev, generator=g) * 0.3
Y = torch.randn((I, F), device=dev, generator=g) * 0.3
lib = _lib.require()
out = (ctypes.c_ulonglong * 16)()
modes = [0]
for mode in modes:
    lib.mirec_topk_debug_mode(mode)
    for mask in (Cui, None):
        E.topk(X, Y, 20, mask); torch.cuda.synchronize()
        lib.mirec_topk_debug_read(out, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); E.topk(X, Y, 20, mask); e1.record(); torch.cuda.synchronize()
        lib.mirec_topk_debug_read(out, 1)
        v = list(out)
        nm, ns = max(v[2], 1), max(v[7], 1)
        print("mode %d mask=%s %.1f ms: per tile: matrix mfma+store %.0f barrier %.0f | selector store %.0f load-issue %.0f selection %.0f barrier %.0f; final stage %.0f cycles per selector"
              % (mode, mask is not None, e0.elapsed_time(e1), v[0] / nm, v[1] / nm, v[8] / ns, (v[4] - v[8]) / ns, v[5] / ns, v[6] / ns,
                 v[9] / max(ns / ((I + 31) // 32), 1)))
lib.mirec_topk_debug_mode(0)
