"""Turn one capture of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/.

usage: python tools/install_profile.py <tag> <previous tag> "<one-line description of the capture>"
Writes profiles/<tag>_bench_kernel_stats.{md,csv}, _bench_line.json, _pmc_traffic.json; carries the notes below
the table over from the previous tag, removes the previous tag's files (but for the L2 and SQ counter files,
which keep the tag of their own capture) and renames the references to them in the design notes at the
repository root.
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
        + ".  (The boxes of this pool differ by a few per cent:\ncapture h, with identical ALS kernels, ran 14.30 ms per iteration "
          "and 34.39 ms per top-20 pass; the L2 counter file is that capture's.)\n\n")
out += old[old.index("Traffic beyond L2 per launch"):].replace(prev + "_", tag + "_")
open(os.path.join(prof, f"{tag}_bench_kernel_stats.md"), "w").write(out)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
open(os.path.join(prof, f"{tag}_bench_line.json"), "w").write(json.dumps(d) + "\n")
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(prof, f"{tag}_pmc_traffic.json"))
# counter files that no capture of this script renews stay under the tag of the capture they were taken with
keep = ("_l2_hit_rates.json", "_sq_counters.json")
for f in glob.glob(os.path.join(prof, f"{prev}_*")):
    if not f.endswith(keep):
        os.remove(f)
p = os.path.join(root, "DESIGN.md")
text = open(p).read()   # (read before the file is opened for writing)
for k in keep:
    text = text.replace(prev + k, "\0KEEP" + k)
open(p, "w").write(text.replace(prev + "_", tag + "_").replace("\0KEEP", prev))
print(out[:1800])
