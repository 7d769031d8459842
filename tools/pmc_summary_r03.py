#!/usr/bin/env python3
"""gpurun_out/prof3 (tools/collect_r03.sh) -> profiles/r03_*:
  r03_bench_kernel_stats_<task>.csv   rocprofv3 --kernel-trace --stats of the default bench command of the task
  r03_bench_<task>.json               the bench line of that run
  r03_pmc_counters.csv                per task / kernel: mean, min, max of every collected counter per dispatch
  r03_pmc_summary.json                per task: HBM-side traffic per dispatch of the step kernel (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 counts
                                      128-byte read requests at 64 B: MI355X_MICROARCH.md), policy steps per dispatch, issue figures -- what bench.py's roofline.traffic reads
"""
import csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof3")
DST = os.path.join(ROOT, "profiles")


def short(name):
    for k in ("k_step", "k_policy_act", "k_roll", "k_extras"):
        if k in name:
            return name.split("(")[0].replace("void ", "").replace("lg::", "")
    return None


summary = {"round": 3, "tasks": {}}
rows_all = []
for tdir in sorted(glob.glob(os.path.join(SRC, "*"))):
    task = os.path.basename(tdir)
    if not os.path.isdir(tdir):
        continue
    line = json.loads([l for l in open(os.path.join(tdir, "stats.json")) if l.startswith("{")][-1])
    json.dump(line, open(os.path.join(DST, f"r03_bench_{task}.json"), "w"), indent=1)
    stats = max(glob.glob(os.path.join(tdir, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    with open(os.path.join(DST, f"r03_bench_kernel_stats_{task}.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        for r in csv.reader(open(stats)):
            r[0] = r[0][:130]
            w.writerow(r)
    per = {}
    for d in sorted(glob.glob(os.path.join(tdir, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            per.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    pl = {}
    for k, cs in sorted(per.items()):
        for c, vals in sorted(cs.items()):
            v = list(vals.values())
            rows_all.append([task, k, c, len(v), f"{statistics.mean(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
            pl.setdefault(k, {})[c] = {"mean": statistics.mean(v), "min": min(v), "max": max(v), "launches": len(v)}
    pmc_line = json.loads([l for l in open(os.path.join(tdir, "pmc_FETCH_SIZE.json")) if l.startswith("{")][-1])
    # the step kernel of the TIMED launches: the k_step entry with the most dispatches that is not the event-timing graph's plain k_step
    cands = [k for k in pl if k.startswith("k_step")]
    main = max(cands, key=lambda k: pl[k]["FETCH_SIZE"]["mean"])       # (the multi-step kernel moves 20 x the bytes of a single step)
    roll = "Lb1EEv" in main and "lg_rollout_policy" in pmc_line["config"]["launch"]
    steps_per_launch = 20 if "lg_rollout_policy" in pmc_line["config"]["launch"] else 1
    f_kib, w_kib = pl[main]["FETCH_SIZE"]["mean"], pl[main]["WRITE_SIZE"]["mean"]
    sq = pl[main]
    nwaves = None
    entry = {"envs_per_gpu": line["config"]["envs_per_gpu"], "workload": line["config"]["workload"],
             "command": f"bench.py --task {task} --steps 40 --warmup 20 --no-graph (eager launches, one counter row per dispatch)",
             "k_step": {"kernel": main, "steps_per_launch": steps_per_launch,
                        "fetch_kib": f_kib, "write_kib": w_kib, "traffic_bytes_per_launch": (2 * f_kib + w_kib) * 1024,
                        "traffic_bytes_per_policy_step": (2 * f_kib + w_kib) * 1024 / steps_per_launch,
                        "algorithmic_bytes_per_policy_step": line["roofline"]["algorithmic_bytes_per_env_step"] * line["config"]["envs_per_gpu"],
                        "kernel_cycles": sq["SQ_BUSY_CYCLES"]["mean"] / 32.0,
                        "mfma_busy_frac_of_busy_cycles": sq["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * sq["SQ_BUSY_CYCLES"]["mean"] / 32.0),
                        "valu_issue_frac": 4.0 * sq["SQ_INSTS_VALU"]["mean"] / (1024.0 * sq["SQ_BUSY_CYCLES"]["mean"] / 32.0),
                        "wait_any_frac_of_wave_cycles": sq["SQ_WAIT_ANY"]["mean"] / sq["SQ_WAVE_CYCLES"]["mean"],
                        "note": "FETCH_SIZE / WRITE_SIZE: KiB per dispatch, separate passes; SQ_BUSY_CYCLES per shader engine (/ 32 = kernel cycles); SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs"},
             "other_kernels": {k: {"fetch_kib": v.get("FETCH_SIZE", {}).get("mean"), "write_kib": v.get("WRITE_SIZE", {}).get("mean"), "launches": v.get("FETCH_SIZE", {}).get("launches")} for k, v in pl.items() if k != main}}
    entry["k_step"]["traffic_over_algorithmic"] = entry["k_step"]["traffic_bytes_per_policy_step"] / entry["k_step"]["algorithmic_bytes_per_policy_step"]
    summary["tasks"][task] = entry
with open(os.path.join(DST, "r03_pmc_counters.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["task", "kernel", "counter", "launches", "mean", "min", "max"])
    w.writerows(rows_all)
json.dump(summary, open(os.path.join(DST, "r03_pmc_summary.json"), "w"), indent=1)
print(json.dumps({t: {k: v for k, v in e["k_step"].items() if k != "note"} for t, e in summary["tasks"].items()}, indent=1))
