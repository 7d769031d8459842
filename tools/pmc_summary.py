#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof) into the committed summaries under profiles/:
  r01_bench_kernel_stats.csv   kernel-trace --stats of the default bench command
  r01_pmc_counters.csv         per-kernel mean/min/max of every collected counter
  r01_pmc_summary.json         the same + HBM traffic per k_step launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
"""
import csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

def short(name):
    for k in ("k_step", "k_policy_act", "k_reset", "k_extras"):
        if k in name:
            return name.split("(")[0].replace("void ", "").replace("lg::", "")
    return None

stats = max(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # gpurun merges: keep the newest run
rows = list(csv.reader(open(stats)))
with open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    for r in rows:
        r[0] = r[0][:110]
        w.writerow(r)
per = {}
for d in sorted(glob.glob(os.path.join(SRC, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in [max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)]:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            per.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
summary = {"round": int(tag[1:]) if tag[1:].isdigit() else tag, "workload": "anymal_c_flat (self-collision on), 4096 envs, 1 x MI355X, eager launches (bench.py --no-graph) so each dispatch has its own counter row",
           "command": "tools/collect_profiles.sh (rocprofv3 --pmc <group> --kernel-trace, one group per pass: FETCH_SIZE | WRITE_SIZE | SQ_*)",
           "per_launch": {}}
with open(os.path.join(DST, f"{tag}_pmc_counters.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "launches", "mean", "min", "max"])
    for k, cs in sorted(per.items()):
        for c, vals in sorted(cs.items()):
            v = list(vals.values())
            w.writerow([k, c, len(v), f"{statistics.mean(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
            summary["per_launch"].setdefault(k, {})[c] = {"mean": statistics.mean(v), "min": min(v), "max": max(v), "launches": len(v)}
main = max((k for k in per if k.startswith("k_step")), key=lambda k: len(next(iter(per[k].values()))))
f_kib = summary["per_launch"][main]["FETCH_SIZE"]["mean"]; w_kib = summary["per_launch"][main]["WRITE_SIZE"]["mean"]
summary["dominant_kernel"] = main
summary["k_step_traffic_bytes"] = {"fetch_raw": f_kib * 1024, "write_raw": w_kib * 1024, "raw_sum": (f_kib + w_kib) * 1024,
                                   "fetch_doubled_sum": (2 * f_kib + w_kib) * 1024,
                                   "note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950 tallies 128-B read requests at 64 B -> fetch doubled (MI355X_MICROARCH.md, HBM section)"}
sq = summary["per_launch"][main]
if "SQ_INSTS_VALU" in sq:
    summary["k_step_issue"] = {"valu_per_wave": sq["SQ_INSTS_VALU"]["mean"] / 1024.0, "note": "1024 waves per launch (256 workgroups x 4 waves)",
                               "wait_any_frac_of_wave_cycles": sq["SQ_WAIT_ANY"]["mean"] / sq["SQ_WAVE_CYCLES"]["mean"],
                               # SQ_BUSY_CYCLES is counted per shader engine (32 on MI355X): / 32 = the kernel's duration in cycles;
                               # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs
                               "kernel_cycles": sq["SQ_BUSY_CYCLES"]["mean"] / 32.0,
                               "mfma_busy_frac_of_busy_cycles": sq["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * sq["SQ_BUSY_CYCLES"]["mean"] / 32.0),
                               "valu_issue_frac": 4.0 * sq["SQ_INSTS_VALU"]["mean"] / (1024.0 * sq["SQ_BUSY_CYCLES"]["mean"] / 32.0)}
json.dump(summary, open(os.path.join(DST, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_launch"}, indent=1))
