#!/bin/bash
# Round-3 evidence for the wide learner (through gpurun from the repo root): rocprofv3 kernel trace of tools/wide_probe.py 235 (one forward +
# backward of the 235-512-256-128-{12,1} actor + critic on 24 576 rows) and the --pmc passes of tools/wide_pmc.sh (FETCH_SIZE | WRITE_SIZE |
# SQ_*, separate passes) -> gpurun_out/r03_wide_mlp_kernels.txt, gpurun_out/r03_wide_mlp_pmc.txt (copied to profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
rm -rf $OUT/wide_trace $OUT/wide_pmc
mkdir -p $OUT/wide_trace
export PYTHONPATH=$ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wide_trace -o wide -- python3 $ROOT/tools/wide_probe.py 235 > $OUT/wide_trace/run.txt 2> $OUT/wide_trace/run.err || exit 1
python3 - "$OUT" <<'PY' || exit 1
import csv, glob, sys
out = sys.argv[1]
trace = glob.glob(out + "/wide_trace/**/*kernel_trace.csv", recursive=True)[0]
stats = glob.glob(out + "/wide_trace/**/*kernel_stats.csv", recursive=True)[0]
run = [l.strip() for l in open(out + "/wide_trace/run.txt") if l.startswith("obs ")]
rows = [r for r in csv.DictReader(open(trace)) if any(t in r["Kernel_Name"] for t in ("k_gemm_wide", "k_wide_", "k_mlp_chain", "k_chain_pack"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = sum(1 for r in rows[-40:] if "k_chain_pack" in r["Kernel_Name"])          # launches per iteration = distance between two packs
idx = [i for i, r in enumerate(rows) if "k_chain_pack" in r["Kernel_Name"]]
seq = rows[idx[-2] if rows[idx[-1]:] and len(rows) - idx[-1] < idx[-1] - idx[-2] else idx[-1]:][: idx[-1] - idx[-2]]
with open(out + "/r03_wide_mlp_kernels.txt", "w") as f:
    f.write("# tools/collect_wide_r03.sh: tools/wide_probe.py 235 under rocprofv3 --kernel-trace --stats: one forward + backward of actor + critic, 24 576 rows\n")
    f.write("\n".join(run) + "\n\n# one iteration, launch by launch (<0/1/2> = GEMM mode FWD/DX/DW; grid x, y)\n")
    t0 = int(seq[0]["Start_Timestamp"])
    for r in seq:
        n = r["Kernel_Name"]; short = n[n.find("lg::") + 4:].split("(")[0][:28]
        f.write("%-28s %7s %4s  start %7.1f us  dur %6.1f us\n" % (short, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    f.write("span of the iteration: %.1f us, sum of kernel durations %.1f us\n" % ((int(seq[-1]["End_Timestamp"]) - t0) / 1e3,
            sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq) / 1e3))
    f.write("\n# kernel table of the whole probe\n")
    for r in list(csv.DictReader(open(stats)))[:9]:
        f.write("%6.2f %%  calls %7s  avg %9.2f us  %s\n" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
print(open(out + "/r03_wide_mlp_kernels.txt").read())
PY
bash $ROOT/tools/wide_pmc.sh || exit 1
python3 - "$OUT" <<'PY'
import ast, sys, re
out = sys.argv[1]
tab = {}
for i in (1, 2, 3):
    for line in open(f"{out}/wide_pmc/p{i}.summary.txt"):
        m = re.match(r"^(\(.*?\)) (\{.*\}) launches (\d+)", line.strip())
        if not m: continue
        key = ast.literal_eval(m.group(1)); vals = ast.literal_eval(m.group(2))
        tab.setdefault((key[0], key[1]), {}).update(vals); tab[(key[0], key[1])]["launches"] = int(m.group(3))
dur = {}
import csv, glob
for f in glob.glob(out + "/wide_pmc/p3/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_BUSY_CYCLES": continue
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
        dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/r03_wide_mlp_pmc.txt", "w") as f:
    f.write("# tools/collect_wide_r03.sh: rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) over tools/wide_probe.py 235; mean per launch.\n"
            "# HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters, the gfx950 correction of profiles/r03_pmc_summary.json); us = duration in the kernel trace of the SQ pass.\n")
    f.write("%-52s %9s %7s %6s %12s %10s %14s\n" % ("kernel (grid threads)", "HBM MB", "us", "TB/s", "VALU insts", "LDS insts", "MFMA busy cyc"))
    total = 0.0
    for (name, grid), v in sorted(tab.items()):
        if "FETCH_SIZE" not in v: continue
        mb = (2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024 / 1e6
        us = sum(dur.get((name, grid), [0])) / max(1, len(dur.get((name, grid), [0])))
        short = name[name.find("lg::") + 4:].split("(")[0][:36]
        f.write("%-52s %9.1f %7.1f %6.2f %12.0f %10.0f %14.0f\n" % (f"{short} ({grid})", mb, us, mb / us if us else 0, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
        per_iter = v["launches"] / 7.0       # WIDE_ITERS=6 timed + 1 warm-up pass
        total += mb * per_iter
    f.write("# HBM traffic of one forward + backward of both nets (launch counts per iteration applied): %.2f GB\n" % (total / 1e3))
print(open(out + "/r03_wide_mlp_pmc.txt").read())
PY
