#!/usr/bin/env python3
"""Distils gpurun_out/prof/ (written by tools/profile_bench.sh) into the summaries kept under profiles/:
  <tag>_bench.json          the bench line
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary (per-kernel calls / total / average ns)
  <tag>_pmc.json            mean counter value per kernel dispatch, per kernel
  pmc_digest.json           per workload ("scene:WxH:limitN:order") and kernel (plus "_build_id": the library build they belong to): HBM bytes per launch and the derived figures
                            bench.py reports (roofline.traffic, roofline.counters); entries of other workloads are kept
usage: tools/profile_digest.py <tag> [directory under gpurun_out/, default prof]"""
import collections, csv, glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, tag = os.path.join(root, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "prof"), os.path.join(root, "profiles"), sys.argv[1]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
bench = json.load(open(os.path.join(src, "bench.json")))
cfg = bench["config"]
order = cfg["traversal_order"].split(" ")[0]
key = f"{cfg['scene']}:{cfg['width']}x{cfg['height']}:limit{cfg['max_recursion_limit']}:{order}"
def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
kernel_ns = {}
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
    for row in csv.DictReader(open(stats[0])):
        kernel_ns[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
pmc = {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())
       if any(t in k for t in ("render_kernel", "render_batch_kernel", "blend_batch_kernel", "spheres_kernel", "traverse_kernel", "shade_kernel", "path_kernel"))}
json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1)
N_SIMD = 256 * 4          # MI355X: 256 compute units of four SIMDs
digest = {}
for k, cs in pmc.items():
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    base = k.split("<")[0]
    targs = [a.strip() for a in k.split("<", 1)[1].rstrip(">").split(",")] if "<" in k else []
    if len(targs) > 1 and targs[1] == "true":
        continue   # (second template argument: the variant with work counters on, which is not the one that is timed)
    m = {c: v["mean"] for c, v in cs.items()}
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 bytes, so it is
    # doubled (exact for wide coalesced reads, an upper bound for this kernel's scattered 64-byte gathers)
    d = {"hbm_bytes_per_launch": int((2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024), "fetch_size_kib_raw": m["FETCH_SIZE"], "write_size_kib_raw": m["WRITE_SIZE"],
         "kernel": k, "source": tag + "_pmc.json"}
    if k in kernel_ns:
        d["kernel_ms"], d["kernel_calls"] = kernel_ns[k][0] * 1e-6, kernel_ns[k][1]
        d["hbm_gbs"] = d["hbm_bytes_per_launch"] / (d["kernel_ms"] * 1e-3) / 1e9
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0     # summed over the 8 XCDs
        d["cycles"] = cyc
        if "kernel_ms" in d:
            d["clock_ghz"] = cyc / (d["kernel_ms"] * 1e6)
        if "SQ_INSTS_VALU" in m:
            d["valu_issue_frac"] = m["SQ_INSTS_VALU"] / (cyc * N_SIMD * 0.5)      # a wave64 VALU instruction occupies its SIMD for 2 cycles
        if "SQ_WAVE_CYCLES" in m:
            d["occupancy_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4.0 / (cyc * N_SIMD)   # SQ_WAVE_CYCLES counts quad-cycles
    if m.get("SQ_ACTIVE_INST_VALU"):
        d["lanes_active_of_64"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
    if m.get("SQ_INSTS_VMEM_RD") and "lanes_active_of_64" in d:
        # per-lane load instructions of one launch, taking the lane occupancy of the VALU instructions for that of the loads; with
        # TCC_HIT + TCC_MISS (requests that reach L2) what tools/experiments/gather_bench.hip measures the load path's ceiling in
        d["vmem_rd_wave_insts"] = m["SQ_INSTS_VMEM_RD"]
        d["lane_loads_per_launch"] = m["SQ_INSTS_VMEM_RD"] * d["lanes_active_of_64"]
    if m.get("TCC_HIT_sum") is not None:
        d["l2_requests_per_launch"] = m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)
    if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m:
        d["wait_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if bench.get("roofline", {}).get("kernel") == base:
        d["bound_one_gpu"] = bench["roofline"]["bound"]      # what bench.py concluded from these counters (carried over to its multi-GPU lines)
    total = lambda e: e.get("kernel_ms", 0.0) * e.get("kernel_calls", 0)
    if base not in digest or total(d) >= total(digest[base]):     # several instantiations of one kernel: the one most of the time goes to
        digest[base] = d
path = os.path.join(dst, "pmc_digest.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
digest["_build_id"] = bench.get("build_id")     # rt_build_id() of the library the counters were collected on: bench.py reports counters_stale when it differs
allw[key] = digest
json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
print(key)
print(json.dumps(digest, indent=1))
