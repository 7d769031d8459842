#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (graph replays of the fused actor+step kernel)
#   2. three --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_*), eager launches so every dispatch gets its own counter row
# Outputs land in gpurun_out/prof/<pass>/ ; tools/pmc_summary.py turns them into profiles/*.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[1/4] kernel trace"; timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 2000 --warmup 100 --no-cpu-baseline --training-iters 0 > "$OUT/stats.json" 2> "$OUT/stats.err" || exit 1
i=2
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
    name=$(echo $grp | cut -d' ' -f1)
    echo "[$i/4] pmc $name"
    timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --training-iters 0 --no-graph --event-steps 20 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || exit 1
    i=$((i + 1))
done
find "$OUT" -name "*.csv" | while read f; do echo "$f $(wc -l < $f)"; done
