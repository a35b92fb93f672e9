"""Condense gpurun_out/prof_<tag>/ (scripts/profile_r02.sh) and gpurun_out/prof3_<tag>/ (scripts/profile_config3.sh)
into profiles/: rocprofv3 kernel stats of the default step, of the fp32-only leg and of BASELINE config 3, the
per-launch PMC averages of the level kernels, and the HBM-traffic figure bench.py quotes.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are collected in
separate --pmc passes; rocprofv3 reports them in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide
coalesced reads at 64 B, so it is doubled; WRITE_SIZE is taken as is.

    python scripts/summarize_prof_r02.py <tag>        # e.g. r02
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
LEVEL_KERNELS = ("mmp_level_lean_kernel", "mmp_level_followup_kernel", "mmp_level_kernel", "screen16_kernel",
                 "mmp_level_tiled16_kernel", "mmp_level_tiled_kernel", "reduce_kernel")
SIMDS = 1024


def counters(path):
    acc, info = defaultdict(lambda: defaultdict(list)), {}
    for row in csv.DictReader(open(path)):
        k = next((k for k in LEVEL_KERNELS if row["Kernel_Name"].startswith(k)), None)
        if k is None:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        info[k] = {f: row[f] for f in ("Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                       "Accum_VGPR_Count", "SGPR_Count", "Grid_Size") if f in row}
    return acc, info


def summarize(src, prefix, groups, command):
    out = {"command": command, "per_kernel": {}}
    totals = defaultdict(float)
    launches = defaultdict(int)
    for g in groups:
        files = glob.glob(os.path.join(src, g, "*counter_collection.csv"))
        if not files:
            continue
        acc, info = counters(files[0])
        for k, d in acc.items():
            e = out["per_kernel"].setdefault(k, {"counters_avg_per_launch": {}, "launches": {}, "pass": {}})
            for name, vals in d.items():
                key = name if g != "pmc_sq_fp32" else name + " (fp32-only run)"
                e["counters_avg_per_launch"][key] = sum(vals) / len(vals)
                e["launches"][key] = len(vals)
                if name in ("FETCH_SIZE", "WRITE_SIZE"):
                    totals[name] += sum(vals)
                    launches[name + ":" + k] = len(vals)
            e["resources"] = info.get(k, {})
    for k, e in out["per_kernel"].items():
        c = e["counters_avg_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            e["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * SIMDS)
    return out, totals, launches


# ---- config 2 (bench.py) -----------------------------------------------------------------------------------------
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
if os.path.isdir(src):
    for sub, name in (("trace", f"{tag}_kernel_stats.csv"), ("trace_fp32", f"{tag}_fp32_leg_kernel_stats.csv")):
        f = glob.glob(os.path.join(src, sub, "*kernel_stats.csv"))
        if f:
            shutil.copy(f[0], os.path.join(dst, name))
    summary, totals, launches = summarize(
        src, tag, ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq_fp32"),
        "rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --no-fp32-leg --no-config5 --steps 10 --warmup 2 "
        "(pmc_sq_fp32: the same with --fp32-only)")
    # busy fraction from wall time where GRBM_GUI_ACTIVE is not in the pass: the kernel-trace average duration
    stats = {}
    f = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
    if f:
        for row in csv.DictReader(open(f[0])):
            stats[row["Name"].split("(")[0]] = float(row["AverageNs"])
    summary["kernel_trace_average_ns"] = stats
    n_calls = launches.get("FETCH_SIZE:reduce_kernel", 0)
    if n_calls and "FETCH_SIZE" in totals and "WRITE_SIZE" in totals:
        fetch = totals["FETCH_SIZE"] * 1024 * 2 / n_calls          # KiB -> B, gfx950 x2 correction; per csc_mmpdf_batch call
        write = totals["WRITE_SIZE"] * 1024 / n_calls
        summary["hbm"] = {"fetch_bytes_per_call_corrected": fetch, "write_bytes_per_call": write,
                          "bytes_per_call": fetch + write, "calls": n_calls}
        json.dump({"bytes_per_launch": (fetch + write) / 5, "bytes_per_call": fetch + write, "fetch_bytes_per_call": fetch,
                   "write_bytes_per_call": write, "profile": tag,
                   "note": "all kernels of one csc_mmpdf_batch call (3200 signals); bytes_per_launch = per tree level "
                           "(call / 5), the unit of bench.py's roofline; FETCH_SIZE KiB x1024 x2 (gfx950 correction), "
                           "WRITE_SIZE KiB x1024; separate --pmc passes"},
                  open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print("config 2:", json.dumps(summary.get("hbm", {})), {k: v.get("mfma_busy_fraction_of_simd_cycles")
                                                            for k, v in summary["per_kernel"].items()})

# ---- config 3 ----------------------------------------------------------------------------------------------------
src3 = os.path.join(ROOT, "gpurun_out", f"prof3_{tag}")
if os.path.isdir(src3):
    f = glob.glob(os.path.join(src3, "trace", "*kernel_stats.csv"))
    if f:
        shutil.copy(f[0], os.path.join(dst, f"{tag}_config3_kernel_stats.csv"))
    summary3, _, _ = summarize(src3, tag, ("pmc_sq", "pmc_lds"),
                               "rocprofv3 --pmc <group> -- python3 scripts/run_config3.py <signals> 2")
    # both passes are needed for the busy fraction (MFMA cycles in pmc_sq, GRBM_GUI_ACTIVE in pmc_lds)
    for k, e in summary3["per_kernel"].items():
        c = e["counters_avg_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            e["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * SIMDS)
            e["mfma_f16_tflop_per_launch"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512 / 1e12
    logs = glob.glob(os.path.join(src3, "trace.log"))
    if logs:
        for line in open(logs[0]):
            if line.startswith("{"):
                summary3["run"] = json.loads(line)
    json.dump(summary3, open(os.path.join(dst, f"{tag}_config3_pmc_summary.json"), "w"), indent=1)
    print("config 3:", {k: v.get("mfma_busy_fraction_of_simd_cycles") for k, v in summary3["per_kernel"].items()},
          summary3.get("run"))

# ---- L2 view and instruction mix (scripts/profile_config3_l2.sh, profile_config3_sq.sh, profile_config2_sq.sh) ----
def per_kernel_totals(path, kernels):
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(path)):
        k = next((k for k in kernels if row["Kernel_Name"].startswith(k + "(") or row["Kernel_Name"] == k), None)
        if k is None:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k][row["Counter_Name"]] += 1
    return {k: {c: {"sum": v, "launches": n[k][c]} for c, v in d.items()} for k, d in acc.items()}


extra = {}
for name, sub, groups, kernels in (
        ("config3_l2", f"prof3l2_{tag}", ("pmc_fetch", "pmc_tcc", "pmc_tcp"), ("screen16_kernel", "mmp_level_tiled16_kernel")),
        ("config3_instruction_mix", f"prof3sq_{tag}", ("pmc_a", "pmc_b"), ("screen16_kernel", "mmp_level_tiled16_kernel")),
        ("config2_instruction_mix", f"prof2sq_{tag}", ("pmc_a", "pmc_b"), ("mmp_level_lean_kernel", "mmp_level_kernel"))):
    d = {}
    for g in groups:
        f = glob.glob(os.path.join(ROOT, "gpurun_out", sub, g, "*counter_collection.csv"))
        if f:
            for k, v in per_kernel_totals(f[0], kernels).items():
                d.setdefault(k, {}).update(v)
    if d:
        extra[name] = d
if extra:
    extra["note"] = ("sums over all launches of the run (config 3: run_config3.py 2048 2 = 3 solves x 5 levels; config 2: bench.py "
                     "--steps 10 --warmup 2 and its set-up calls); SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* are in units of 4 clocks; "
                     "FETCH_SIZE in KiB before the gfx950 x2 correction")
    json.dump(extra, open(os.path.join(dst, f"{tag}_pmc_extra.json"), "w"), indent=1)
    print("extra:", list(extra))
