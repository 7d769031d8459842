#!/bin/bash
# Round-3 rocprofv3 evidence for bench.py, per BASELINE config measured on one GPU (run through gpurun from the repo root):
#   per task: 1. --kernel-trace --stats of the default bench command; 2. three --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*) of the same
#   launch mode with eager launches (--no-graph: every dispatch gets its own counter row; for anymal_c_flat that is still lg_rollout_policy,
#   20 policy steps per dispatch).  Outputs: gpurun_out/prof3/<task>/...; tools/pmc_summary_r03.py turns them into profiles/r03_*.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof3
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for task in ${LG_TASKS:-anymal_c_flat anymal_c_rough cassie}; do
    mkdir -p "$OUT/$task"
    echo "[$task] kernel trace"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$task/stats" -- python3 "$ROOT/bench.py" --task $task --steps 2000 --warmup 100 --no-cpu-baseline --training-iters 0 > "$OUT/$task/stats.json" 2> "$OUT/$task/stats.err" || { tail -5 "$OUT/$task/stats.err"; exit 1; }
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
        name=$(echo $grp | cut -d' ' -f1)
        echo "[$task] pmc $name"
        timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$task/pmc_$name" -- python3 "$ROOT/bench.py" --task $task --steps 40 --warmup 20 --no-cpu-baseline --training-iters 0 --no-graph --event-steps 20 > "$OUT/$task/pmc_$name.json" 2> "$OUT/$task/pmc_$name.err" || { tail -5 "$OUT/$task/pmc_$name.err"; exit 1; }
    done
done
find "$OUT" -name "*.csv" | while read f; do echo "$f $(wc -l < $f)"; done
