"""In-tree build of ``libmirec.so`` (hipcc, gfx950 only).

hipcc cross-compiles without a GPU, so this runs in a GPU-less container; the
shared object is written next to this file and travels with the tree.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(HERE, "libmirec.so")
SOURCES = ["core.hip", "gram.hip", "als_sweep.hip", "als_short.hip", "plan_sort.hip", "topk.hip", "topk_small.hip", "topk_order.hip", "topk_inst_0.hip", "topk_inst_1.hip",
           "topk_inst_2.hip", "topk_inst_3.hip", "topk_inst_4.hip", "topk_inst_5.hip", "lightfm.hip", "lightfm_train.hip", "transpose.hip", "metrics.hip", "build_matrix.hip"]
HEADERS = sorted(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")) + [os.path.join(ROOT, "include", "mirec.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("MIREC_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source to an object and link them; returns the .so path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for src in sources:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{res.stdout}\n{res.stderr}")
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MAX_JOBS", "8")))) as pool:
            list(pool.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in sources]
    if force or jobs or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
