#!/bin/bash
# Training curves for profiles/rNN_training.txt (run through gpurun from the repo root): tools/train_probe.py per task,
# every 50th iteration line + the summary lines.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/training.txt
: > "$OUT"
run() {   # name iterations task
    echo "== $1 ($3, $2 iterations)" >> "$OUT"
    timeout -k 10 400 python3 "$ROOT/tools/train_probe.py" "$2" "$3" 2>&1 | awk '/^it /{split($2,a,"/"); if (a[1]==0 || (a[1]+1)%50==0) print; next} /^total|^episode terms|^eval/{print}' | cut -c1-400 >> "$OUT" || exit 1
    echo >> "$OUT"
}
run train_flat 400 anymal_c_flat && run train_rough 300 anymal_c_rough && run train_cassie 300 cassie && run train_anymal_b 400 anymal_b
cat "$OUT"
