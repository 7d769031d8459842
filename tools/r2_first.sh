#!/bin/bash
# round-2 first GPU pass: full-size parity tests, then bench lines + rocprofv3 kernel tables for configs 3 and 5
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_full_size.py -q -m gpu > "$OUT/fullsize.log" 2>&1
rc=$?
echo "fullsize rc=$rc"; tail -30 "$OUT/fullsize.log"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --task anymal_c_rough --steps 2000 --warmup 100 > "$OUT/bench_rough.json" 2> "$OUT/bench_rough.err" || exit 1
timeout -k 10 200 python bench.py --task cassie --num-envs 8192 --steps 2000 --warmup 100 > "$OUT/bench_cassie.json" 2> "$OUT/bench_cassie.err" || exit 1
cat "$OUT/bench_rough.json" "$OUT/bench_cassie.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_rough" -- python3 "$ROOT/bench.py" --task anymal_c_rough --steps 2000 --warmup 100 --no-cpu-baseline > "$OUT/stats_rough.json" 2> "$OUT/stats_rough.err" || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_cassie" -- python3 "$ROOT/bench.py" --task cassie --num-envs 8192 --steps 2000 --warmup 100 --no-cpu-baseline > "$OUT/stats_cassie.json" 2> "$OUT/stats_cassie.err" || exit 1
find "$OUT" -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 "$f"; done
