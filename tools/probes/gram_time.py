"""Time and accuracy of mirec_gram at the benchmark's shapes.

usage: gram_time.py [libmirec.so]   (a variant built by tools/probes/build_variant.py, or the regular build)
"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from movie_recommender_system_amd import _lib
if len(sys.argv) > 1:
    _lib.load(sys.argv[1])
from movie_recommender_system_amd import engine as E  # noqa: E402
dev = torch.device("cuda", 0)
for n, f in [(1_000_000, 128), (27_000, 128), (1_000_000, 64)]:
    X = torch.rand((n, f), device=dev)
    for _ in range(2): G = E.gram(X, 0.01)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): G = E.gram(X, 0.01)
    b.record(); torch.cuda.synchronize()
    ref = (X.double().T @ X.double()).float()
    print(n, f, "%.3f ms" % (a.elapsed_time(b) / 10), "max rel err %.2e" % ((G - ref - 0.01 * torch.eye(f, device=dev)).abs().max() / ref.abs().max()).item(), bool((G == G.T).all()))
