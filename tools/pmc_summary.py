"""Fold rocprofv3 `--pmc ... --output-format csv` passes into per-kernel, per-launch averages.

    python tools/pmc_summary.py out.json dir1 dir2 ...

Directories are named <tag>_<COUNTER>; passes with the same tag are merged.

Each directory holds one pass (`*_counter_collection.csv`).  HBM-side traffic per
launch is (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: both counters are in KiB,
and on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
            per_dispatch = collections.defaultdict(float)
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"]
                if "mirec::" not in name:
                    continue
                name = name.split("(")[0].replace("void ", "")
                per_dispatch[(os.path.basename(d.rstrip("/")), name, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (tag, name, _, ctr), v in per_dispatch.items():
                acc[(tag.split("_")[0], name)][ctr].append(v)
    res = {}
    for (tag, name), ctrs in sorted(acc.items()):
        e = {c: sum(v) / len(v) for c, v in ctrs.items()}
        e["launches"] = {c: len(v) for c, v in ctrs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        res.setdefault(tag, {})[name] = e
    res["workload"] = {"users": 1000000, "items": 27000, "mean_degree": 50.0, "seed": 42, "factors": 128, "topk": 20}
    res["how"] = ("one rocprofv3 --pmc pass per counter over tools/microbench.py (tools/pmc_passes.sh); "
                  "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for tag, ks in res.items():
        if isinstance(ks, dict):
            for k, e in ks.items():
                if isinstance(e, dict) and "hbm_bytes_per_launch" in e:
                    print(tag, k, "%.3f GB per launch" % (e["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
