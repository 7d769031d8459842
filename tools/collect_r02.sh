#!/bin/bash
# Round-2 evidence beyond tools/collect_profiles.sh (run through gpurun from the repo root; outputs under gpurun_out/r02/):
#   bench lines + rocprofv3 kernel tables of BASELINE configs 3 and 5, the wide learner's per-launch timeline, per-workgroup
#   section profiles of k_step, training curves.  tools/collect_r02.py copies the summaries into profiles/.
# Before the gpurun call, HERE: python tools/profile_sections.py build && python tools/profile_sections.py build light  (the two -DLG_PROFILE libraries
# travel with the snapshot; a stale one profiles the OLD kernel source).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for task in anymal_c_rough cassie; do
    echo "[bench + kernel table] $task"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$task" -- python3 "$ROOT/bench.py" --task $task --training-iters 0 --no-cpu-baseline > "$OUT/bench_${task}_profiled.json" 2> "$OUT/bench_$task.err" || exit 1
    cp "$(find "$OUT/stats_$task" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_${task}_kernel_stats.csv"; rm -rf "$OUT/stats_$task"
    timeout -k 10 300 python3 "$ROOT/bench.py" --task $task > "$OUT/bench_$task.json" 2>> "$OUT/bench_$task.err" || exit 1
done
echo "[wide learner timeline]"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/wide" -- python3 "$ROOT/tools/wide_probe.py" 235 > "$OUT/wide_probe.txt" 2> "$OUT/wide.err" || exit 1
python3 "$ROOT/tools/gemm_timeline.py" "$(find "$OUT/wide" -name '*kernel_trace.csv' | head -1)" > "$OUT/wide_timeline.txt"
cp "$(find "$OUT/wide" -name '*kernel_stats.csv' | head -1)" "$OUT/wide_kernel_stats.csv"; rm -rf "$OUT/wide"
cd "$ROOT"
echo "[section profiles]"
for t in "anymal_c_flat 4096" "anymal_c_rough 4096" "cassie 8192"; do
    set -- $t
    timeout -k 10 200 python3 tools/profile_sections.py run $1 $2 > "$OUT/sections_$1.txt" 2>&1 || exit 1
    timeout -k 10 200 python3 tools/profile_sections.py run $1 $2 light > "$OUT/light_$1.txt" 2>&1 || exit 1      # needs `build light` too
done
echo "[training curves]"
for t in "anymal_c_flat 300" "anymal_c_rough 300" "cassie 300"; do
    set -- $t
    timeout -k 10 400 python3 tools/train_probe.py $2 $1 > "$OUT/train_$1.log" 2>&1 || exit 1
    grep "^it " "$OUT/train_$1.log" | tail -1 | cut -c1-200
done
ls "$OUT"
