"""Turn one capture of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/.

usage: python tools/install_profile.py <tag> <previous tag> "<one-line description of the capture>"
Writes profiles/<tag>_bench_kernel_stats.{md,csv}, _bench_line.json, _pmc_traffic.json, with the traffic table made
from the counter file.  A previous tag of the SAME round (same rNN prefix) is replaced: its files are removed (but
for the L2 and SQ counter files, which keep the tag of their own capture) and the references to them in the design
notes at the repository root renamed; an earlier round's files stay.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

tag, prev, what = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
prof = os.path.join(root, "profiles")
old = open(os.path.join(prof, f"{prev}_bench_kernel_stats.md")).read()
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
head = old.split("| kernel |")[0].replace(prev, tag)
head = re.sub(r"^# .*$", f"# rocprofv3 --kernel-trace --stats, {what}", head, count=1, flags=re.M)
name = lambda n: re.sub(r"\(.*$", "", re.sub(r"^void ", "", n))
out = head + "| kernel | calls | avg ms | max ms | total ms | % |\n|---|---|---|---|---|---|\n"
for r in rows[:24]:
    out += (f"| {name(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {float(r['MaxNs']) / 1e6:.4f} | "
            f"{float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |\n")
d = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().split("\n")[-1])
k, e = d["extra"]["kernel_ms_per_launch"], d["extra"]
out += (f"\nbench.py's live HIP-event figures without the profiler (`{tag}_bench_line.json`): als_short_kernel "
        f"{k['user_short']:.3f} ms/launch,\nals_gram_cg_kernel {k['item_main']:.3f} (item side) and {k['user_main']:.3f} (user side), "
        f"als_finalize_kernel {k['item_finalize']:.3f}, user-side Gram {k['gram_users']:.3f};\n{d['ms_per_step']:.2f} ms per iteration; "
        f"top-20 pass over all users {e['top20_pass_ms']:.2f} ms (wall clock around the launch, workspace allocation\nand plane split "
        f"included); WARP epoch {e['lightfm_warp_examples_per_s_incl_transfers'] / 1e6:.1f} M examples/s including its transfers"
        + (f" ({e['lightfm_warp_first_epoch_examples_per_s_incl_transfers'] / 1e6:.1f} M/s for a fresh model's first epoch)"
           if "lightfm_warp_first_epoch_examples_per_s_incl_transfers" in e else "")
        + f"; the sweep order of the top-20 pass built in {e.get('top20_order_build_ms', float('nan')):.1f} ms, once per matrix."
          "  (The boxes of this pool differ by up to a tenth in kernel time.)\n\n")
t = json.load(open(os.path.join(src, "pmc_traffic.json")))
gb = lambda side, kern: next((v["hbm_bytes_per_launch"] / 1e9 for n, v in t.get(side, {}).items() if kern in n), float("nan"))
rl = {r["side"] + ":" + r["kernel"].split("<")[0]: r for r in e.get("half_sweep_rooflines", [])}
alg = lambda key: rl[key]["bytes_per_launch"] / 1e9 if key in rl else float("nan")
out += (f"Traffic beyond L2 per launch (`{tag}_pmc_traffic.json`; TCC FETCH_SIZE and WRITE_SIZE, one counter per pass, in units of\n"
        "1 KiB with the read counter doubled as the gfx950 correction prescribes):\n\n"
        "| kernel | beyond-L2 GB per launch | algorithmic GB per launch |\n|---|---|---|\n"
        f"| als_short_mfma_kernel (user half-sweep) | {gb('user', 'als_short_mfma_kernel'):.2f} | {alg('user:als_short_mfma_kernel'):.2f} |\n"
        f"| als_gram_cg_kernel (item half-sweep) | {gb('item', 'als_gram_cg_kernel'):.2f} | {alg('item:als_gram_cg_kernel'):.2f} |\n"
        f"| topk_planes_kernel (all-users top-20, items by descending interactions, trained factors) | {gb('topk', 'topk_planes_kernel'):.2f} | "
        "0.93 (queries, item planes once, liked items, results) |\n\n"
        "The user half-sweep's gathered table (27 000 x 128 fp32 = 13.8 MB) exceeds the 4 MiB L2 of an XCD and is served by the\n"
        "Infinity Cache: its traffic beyond L2 is gather re-reads, not HBM traffic.  The top-K sweep reads the hi plane of the items\n"
        "(6.9 MB) once per workgroup of 256 queries through L2; what goes beyond L2 is the queries, the results, the filter lists and\n"
        "the part of the final stage's item-row gathers that misses.\n")
open(os.path.join(prof, f"{tag}_bench_kernel_stats.md"), "w").write(out)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
open(os.path.join(prof, f"{tag}_bench_line.json"), "w").write(json.dumps(d) + "\n")
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(prof, f"{tag}_pmc_traffic.json"))
# counter files that no capture of this script renews stay under the tag of the capture they were taken with
keep = ("_l2_hit_rates.json", "_sq_counters.json")
if prev[:3] == tag[:3] and prev != tag:
    for f in glob.glob(os.path.join(prof, f"{prev}_*")):
        if not f.endswith(keep):
            os.remove(f)
    p = os.path.join(root, "DESIGN.md")
    text = open(p).read()   # (read before the file is opened for writing)
    for k in keep:
        text = text.replace(prev + k, "\0KEEP" + k)
    open(p, "w").write(text.replace(prev + "_", tag + "_").replace("\0KEEP", prev))
print(out[:1800])
