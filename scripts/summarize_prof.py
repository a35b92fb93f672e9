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

KERNEL = "mmp_level_kernel"
summary = {"kernel": KERNEL, "command": "rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --no-fp32-leg --steps 10 --warmup 2",
           "counters_avg_per_launch": {}, "by_grid": {}}
for group in ("pmc_fetch", "pmc_write", "pmc_sq"):
    path = os.path.join(src, group, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc, by_grid = defaultdict(list), defaultdict(lambda: defaultdict(list))
    info = {}
    for row in csv.DictReader(open(path)):
        if not row["Kernel_Name"].startswith(KERNEL):
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        by_grid[row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        info = {k: row[k] for k in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                    "Accum_VGPR_Count", "SGPR_Count")}
    for name, vals in acc.items():
        summary["counters_avg_per_launch"][name] = sum(vals) / len(vals)
        summary.setdefault("launches", {})[name] = len(vals)
    for g, d in by_grid.items():
        for name, vals in d.items():
            summary["by_grid"].setdefault(g, {})[name] = sum(vals) / len(vals)
    summary["resources"] = info
c = summary["counters_avg_per_launch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fetch = c["FETCH_SIZE"] * 1024 * 2          # KiB -> B, gfx950 x2 correction
    write = c["WRITE_SIZE"] * 1024
    summary["hbm"] = {"fetch_bytes_per_launch_corrected": fetch, "write_bytes_per_launch": write,
                      "bytes_per_launch": fetch + write}
    json.dump({"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "profile": tag,
               "note": "avg over the 5 level launches of one csc_mmpdf_batch (3200 signals); FETCH_SIZE KiB x1024 x2 "
                       "(gfx950 correction), WRITE_SIZE KiB x1024; separate --pmc passes"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
    summary["mfma_busy_over_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
