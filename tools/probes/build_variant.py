"""Build a variant of libmirec.so: the named sources are recompiled with extra flags, the other objects are
those of the regular build.

usage: build_variant.py <name> <source.hip>[,<source.hip>...] [extra hipcc flags ...]
Writes movie-recommender-system_amd/build/variants/libmirec_<name>.so (the directory travels to the GPU box,
it is not under version control) and prints the path; tools/probes/*.py take such a path.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movie_recommender_system_amd import build as B  # noqa: E402


def main():
    name, sources, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build_lib()
    out_dir = os.path.join(B.OBJ_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        o = os.path.join(B.OBJ_DIR, src.replace(".hip", ".o"))
        if src in sources:
            o = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
            cmd = [B._hipcc()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", o]
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise SystemExit(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
        objs.append(o)
    lib = os.path.join(out_dir, f"libmirec_{name}.so")
    res = subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs,
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit(f"link failed:\n{res.stdout}\n{res.stderr}")
    print(lib)


if __name__ == "__main__":
    main()
