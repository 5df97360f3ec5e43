"""Prints VGPR / AGPR / SGPR / LDS / scratch per kernel of every HIP source (compiles to assembly only).

    python tools/kernel_resources.py [source.hip ...]
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "movie-recommender-system_amd", "csrc")


def one(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, "-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
    text = open(out).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", text, flags=re.S):
        blk = m.group(0)
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void mirec::", "")
        rows.append((name, g("vgpr_count"), m.group(1), g("sgpr_count"), g("group_segment_fixed_size"),
                     g("private_segment_fixed_size"), g("vgpr_spill_count")))
    return rows


if __name__ == "__main__":
    srcs = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, srcs))
    print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s} {'spill':>5s}")
    for rows in res:
        for r in rows:
            print(f"{r[0][:58]:58s} {r[1]:>5s} {r[2]:>5s} {r[3]:>5s} {r[4]:>7s} {r[5]:>7s} {r[6]:>5s}")
