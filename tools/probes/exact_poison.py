"""Does the exact half-sweep read a vector register it never wrote?  Before each sweep every SIMD's register file is
filled with a bit pattern (tools/probes/poison_registers.hip); a result that changes with the pattern says yes.

usage: exact_poison.py [libmirec.so] ; env F (64), USERS (250), ITEMS (700)
"""
import ctypes
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movie_recommender_system_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E, synth  # noqa: E402

poison = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "bin", "libpoison.so"))
dev = torch.device("cuda", 0)
f = int(os.environ.get("F", 64))
U, I = int(os.environ.get("USERS", 250)), int(os.environ.get("ITEMS", 700))
cg = os.environ.get("CG", "0") == "1"
M = synth.interactions(21, U, I, 30.0)
rs = np.random.RandomState(42)
F = E.factor_width(f)
X0 = E.pad_factors(rs.rand(U, f).astype(np.float32) * 0.01, F, dev)
Y0 = E.pad_factors(rs.rand(I, f).astype(np.float32) * 0.01, F, dev)
C = E.DeviceCSR.from_scipy(M, dev, scale=40.0)
plan = E.SweepPlan(C, F, E.auto_chunk(C.nnz), short_max=0, conf_max=40.0)
G = E.gram(Y0, 0.01)
ref = None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for pat in (0x00000000, 0x00000000, 0x7fc00000, 0x3f800000, 0xff800000, 0x00000000, 0x12345678, 0x7fc00000):
    for _ in range(3):
        assert poison.mirec_probe_poison(ctypes.c_uint32(pat), 8192, stream) == 0
    X = X0.clone()
    torch.cuda.synchronize()
    assert poison.mirec_probe_poison(ctypes.c_uint32(pat), 8192, stream) == 0
    plan.sweep(X, Y0, G, 3 if cg else -1)
    torch.cuda.synchronize()
    x = X.cpu().numpy()
    if ref is None:
        ref = x
        continue
    d = np.abs(x - ref).max(axis=1)
    bad = np.nonzero((d > 0) | ~np.isfinite(d))[0]
    print(f"pattern {pat:#010x}: {len(bad)} of {U} rows differ from the first run (pattern 0), non-finite rows {int((~np.isfinite(x).all(axis=1)).sum())}")
