#!/usr/bin/env python3
"""Distils gpurun_out/prof/ (written by tools/profile_bench.sh) into the summaries kept under profiles/:
  <tag>_bench.json          the bench line
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average ns)
  <tag>_pmc.json            mean counter value per kernel dispatch, per kernel
  hbm_traffic.json          HBM bytes per launch of the dominant kernels, read by bench.py for roofline.traffic
usage: tools/profile_digest.py <tag>"""
import collections, csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, tag = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles"), sys.argv[1]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
pmc = {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())
       if any(t in k for t in ("render_kernel", "render_batch_kernel", "blend_batch_kernel", "spheres_kernel", "traverse_kernel", "shade_kernel", "path_kernel"))}
json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
traffic = {}
for k, cs in pmc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        base = k.split("<")[0]
        if ", true" in k and any(o.split("<")[0] == base and ", true" not in o for o in pmc):
            continue   # the variant with work counters on is not the one that is timed
        # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so it is doubled
        # (exact for wide coalesced reads, an upper bound for this kernel's scattered 64-byte gathers)
        traffic[base] = {"hbm_bytes_per_launch": int((2.0 * cs["FETCH_SIZE"]["mean"] + cs["WRITE_SIZE"]["mean"]) * 1024),
                         "fetch_size_kib_raw": cs["FETCH_SIZE"]["mean"], "write_size_kib_raw": cs["WRITE_SIZE"]["mean"], "source": tag + "_pmc.json"}
json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
for k, cs in pmc.items():
    print(k, {c: round(v["mean"], 1) for c, v in cs.items()})
