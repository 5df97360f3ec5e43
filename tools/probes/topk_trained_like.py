"""Top-20 timing on trained-like factors (a popularity component shared by all users: dense score ties near the top),
with the share of queries the plane kernel hands over to the fp32 kernel."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from movie_recommender_system_amd import engine as E, synth, _lib
dev = torch.device("cuda", 0)
U, I, F = 1_000_000, 27_000, 128
indptr, indices, data = synth.interactions_torch(42, U, I, 50.0, dev)
mask = E.DeviceCSR(indptr, indices, data, (U, I))
g = torch.Generator(device=dev); g.manual_seed(5)
pop = torch.rand((I, 1), device=dev, generator=g) ** 4
X = torch.rand((U, F), device=dev, generator=g) * 0.3
Y = (torch.rand((I, F), device=dev, generator=g) * 0.1 + pop * 0.2).contiguous()
lib = _lib.require()
def run(ws):
    E.topk(X, Y, 20, mask, workspace=ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = E.topk(X, Y, 20, mask, workspace=ws); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
print(f"trained-like factors: plane kernel {run(True):.2f} ms, fp32 kernel {run(False):.2f} ms")
# hand-over count: read the list length from a workspace of our own
nb = int(lib.mirec_topk_workspace_bytes(U, I, F, 20))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
ids = torch.empty((U, 20), dtype=torch.int32, device=dev); sc = torch.empty((U, 20), dtype=torch.float32, device=dev)
_lib.check(lib.mirec_topk(E._ptr(X), U, E._ptr(Y), I, F, E._ptr(mask.indptr), E._ptr(mask.indices), None, None, 20,
                          E._ptr(ids), E._ptr(sc), E._ptr(ws), nb, E._stream_ptr()), "mirec_topk")
torch.cuda.synchronize()
print("queries handed over to the fp32 kernel:", int(ws[4:8].view(torch.int32).item()), "of", U)
