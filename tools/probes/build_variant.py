"""Build a variant of libmirec.so: the named sources are recompiled with extra flags, the other objects are
those of the regular build.

usage: build_variant.py <name> <source.hip>[,<source.hip>...] [extra hipcc flags ...]
Writes movie-recommender-system_amd/build/variants/libmirec_<name>.so (the directory travels to the GPU box,
it is not under version control) and prints the path; tools/probes/*.py take such a path.

The shipped kernel sources hold no probe or timing code.  The builds that leave a part of a kernel out
(-DMIREC_PROBE_*), count cycles per phase (-DMIREC_DEBUG_TIMING) or force the vector-ALU short kernel
(-DMIREC_SHORT_VALU_ONLY) live in tools/probes/patches/*.patch: when one of those flags is given, the sources are
copied to a scratch directory, the patches applied there, and the named units compiled from the copy.  A patch that
no longer applies says so: it has to be refreshed against the kernel it instruments.
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from movie_recommender_system_amd import build as B  # noqa: E402


PATCHES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "patches")
INSTRUMENTED = ("-DMIREC_PROBE_", "-DMIREC_DEBUG_TIMING", "-DMIREC_SHORT_VALU_ONLY")


def main():
    name, sources, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build_lib()
    out_dir = os.path.join(B.OBJ_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    csrc = B.CSRC
    if any(f.startswith(INSTRUMENTED) for f in flags):
        csrc = os.path.join(tempfile.mkdtemp(prefix="mirec_probe_"), "csrc")
        shutil.copytree(B.CSRC, csrc)
        for patch in sorted(os.listdir(PATCHES)):
            res = subprocess.run(["patch", "-p1", "-d", csrc, "-i", os.path.join(PATCHES, patch)],
                                 capture_output=True, text=True)
            if res.returncode != 0:
                raise SystemExit(f"{patch} does not apply to the current sources:\n{res.stdout}\n{res.stderr}")
    objs = []
    for src in B.SOURCES:
        o = os.path.join(B.OBJ_DIR, src.replace(".hip", ".o"))
        if src in sources:
            o = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
            cmd = [B._hipcc()] + B.FLAGS + flags + ["-c", os.path.join(csrc, src), "-o", o]
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
