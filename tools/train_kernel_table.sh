#!/bin/bash
# rocprofv3 kernel table of a short training run of one task (GPU box): tools/train_kernel_table.sh <task> <iters> <outdir>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TASK=${1:-anymal_c_rough}; IT=${2:-30}; OUT=$ROOT/gpurun_out/${3:-r2b}/train_prof_$TASK
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/train_probe.py" $IT $TASK > "$OUT/train.log" 2> "$OUT/train.err" || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/training_kernel_stats.csv"
rm -rf "$OUT/stats"
grep "^it " "$OUT/train.log" | tail -2 | cut -c1-160
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/training_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per iteration: %.2f ms" % (tot / 1e6 / $IT))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%6.2f %%  %8.3f ms/iter  calls/iter %7.1f  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6 / $IT, float(r["Calls"]) / $IT, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
