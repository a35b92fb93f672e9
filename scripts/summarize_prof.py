"""Condense gpurun_out/prof_<tag>/ (scripts/profile_r01.sh) into profiles/: the rocprofv3 kernel stats,
the per-launch PMC averages of mmp_level_kernel and the HBM-traffic figure bench.py reports.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes; rocprofv3 reports them in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is taken as is."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))

KERNELS = ("mmp_level_lean_kernel", "mmp_level_followup_kernel", "mmp_level_kernel")   # levels 1-4 (one launch), its fp32 follow-up / level 0
summary = {"kernels": KERNELS,
           "command": "rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2",
           "per_kernel": {}}
totals = defaultdict(float)
n_lean = 0
for group in ("pmc_fetch", "pmc_write", "pmc_sq"):
    path = os.path.join(src, group, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = {k: defaultdict(list) for k in KERNELS}
    info = {}
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        k = next((k for k in KERNELS if name.startswith(k)), None)
        if k is None:
            continue
        key = k
        acc.setdefault(key, defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            totals[row["Counter_Name"]] += float(row["Counter_Value"])
        if k == "mmp_level_lean_kernel" and row["Counter_Name"] == "FETCH_SIZE":
            n_lean += 1
        info[key] = {f: row[f] for f in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                         "Accum_VGPR_Count", "SGPR_Count", "Grid_Size")}
    for key, d in acc.items():
        for name, vals in d.items():
            e = summary["per_kernel"].setdefault(key, {"counters_avg_per_launch": {}, "launches": {}})
            e["counters_avg_per_launch"][name] = sum(vals) / len(vals)
            e["launches"][name] = len(vals)
            e["resources"] = info.get(key, {})
if n_lean and "FETCH_SIZE" in totals and "WRITE_SIZE" in totals:
    levels = n_lean * 5                          # per step: level 0 + ONE lean launch that walks levels 1-4
    fetch = totals["FETCH_SIZE"] * 1024 * 2 / levels          # KiB -> B, gfx950 x2 correction
    write = totals["WRITE_SIZE"] * 1024 / levels
    summary["hbm"] = {"fetch_bytes_per_level_corrected": fetch, "write_bytes_per_level": write,
                      "bytes_per_level": fetch + write, "levels": levels}
    json.dump({"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "profile": tag,
               "note": "level kernels of one csc_mmpdf_batch (3200 signals) / its 5 tree levels; FETCH_SIZE KiB x1024 x2 (gfx950 correction), WRITE_SIZE KiB x1024; separate --pmc passes"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
for key, e in summary["per_kernel"].items():
    c = e["counters_avg_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        # MFMA-busy cycles are summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines' sequencers
        e["mfma_busy_fraction_of_simd_cycles"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["SQ_BUSY_CYCLES"] / 32)
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
