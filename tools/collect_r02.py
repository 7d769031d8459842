#!/usr/bin/env python3
"""gpurun_out/r02/ (tools/collect_r02.sh) -> profiles/r02_*: bench lines + kernel tables of configs 3 and 5, the wide learner's
timeline, per-workgroup section profiles of k_step, training curves."""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")

def top_kernels(path, n=12):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = []
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
        out.append("%6.2f %%  calls %7d  avg %9.2f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:150]))
    return out

for task in ("anymal_c_rough", "cassie"):
    shutil.copy(os.path.join(SRC, f"bench_{task}.json"), os.path.join(DST, f"r02_bench_{task}.json"))
    shutil.copy(os.path.join(SRC, f"bench_{task}_kernel_stats.csv"), os.path.join(DST, f"r02_bench_{task}_kernel_stats.csv"))

with open(os.path.join(DST, "r02_wide_mlp_kernels.txt"), "w") as f:
    f.write("# tools/wide_probe.py 235 under rocprofv3 --kernel-trace --stats: one forward + backward of actor + critic, 24 576 rows\n")
    f.write(open(os.path.join(SRC, "wide_probe.txt")).read())
    f.write("\n# last iteration, launch by launch (first column: r = prep / pack / reduce, 0/1/2 = GEMM mode FWD/DX/DW, chain = 1 with grid.y = nets)\n")
    f.write(open(os.path.join(SRC, "wide_timeline.txt")).read())
    f.write("\n# kernel table of the whole probe\n" + "\n".join(top_kernels(os.path.join(SRC, "wide_kernel_stats.csv"))) + "\n")

with open(os.path.join(DST, "r02_sections.txt"), "w") as f:
    f.write("# tools/profile_sections.py (-DLG_PROFILE build, s_memtime per section on lane 0 of every workgroup, eager env.step with N(0,1) actions)\n")
    for task in ("anymal_c_flat", "anymal_c_rough", "cassie"):
        txt = open(os.path.join(SRC, f"sections_{task}.txt")).read()
        i = txt.find(task + " N=")
        f.write("\n" + re.sub(r"np\.float64\(([^)]*)\)", r"\1", txt[i:] if i >= 0 else txt))
    f.write("\n# ---- the LIGHT build (-DLG_PROFILE_LIGHT: each workgroup's start / end clock and the fallen-robot events only): the spread of the PRODUCT kernel's workgroups\n")
    for task in ("anymal_c_flat", "anymal_c_rough", "cassie"):
        path = os.path.join(SRC, f"light_{task}.txt")
        if os.path.exists(path):
            txt = open(path).read()
            i = txt.find(task + " N=")
            f.write("\n" + re.sub(r"np\.float64\(([^)]*)\)", r"\1", txt[i:] if i >= 0 else txt))

with open(os.path.join(DST, "r02_training.txt"), "w") as f:
    f.write("# tools/train_probe.py <iterations> <task>: bundled PPO runner, 4096 envs, registered configs (rough tasks: 'trimesh' faces, terrain curriculum)\n")
    for task in ("anymal_c_flat", "anymal_c_rough", "cassie"):
        f.write(f"\n== {task}\n")
        for line in open(os.path.join(SRC, f"train_{task}.log")):
            m = re.match(r"it (\d+)/", line)
            if m and (int(m.group(1)) == 0 or (int(m.group(1)) + 1) % 50 == 0):
                f.write(line[:220].rstrip() + "\n")
            elif line.startswith(("total", "eval")):
                f.write(line[:300].rstrip() + "\n")
print(sorted(x for x in os.listdir(DST) if x.startswith("r02")))
print("note: the longer runs at the end of r02_training.txt (tools/train_probe.py 1500 anymal_c_flat / 1000 anymal_c_rough) are appended by hand")
