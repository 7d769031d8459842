#!/bin/bash
# Round-3 training curves for profiles/r03_training.txt (through gpurun from the repo root): tools/train_probe.py per task with the
# reference's PPO defaults at the reference's iteration counts (300 flat, 1500 the others; registered terrain: 'trimesh' faces), every 100th line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/training_r03.txt
: > "$OUT"
run() {   # iterations task
    echo "== $2, $1 iterations" >> "$OUT"
    timeout -k 10 500 python3 "$ROOT/tools/train_probe.py" "$1" "$2" 2>&1 | awk -v n="$1" '/^it /{split($2,a,"/"); if (a[1]==0 || (a[1]+1)%100==0 || a[1]+1==n) print; next} /^total|^episode terms|^eval/{print}' | cut -c1-330 >> "$OUT" || exit 1
    echo >> "$OUT"
}
run 300 anymal_c_flat && run 1500 anymal_c_rough && run 1500 cassie && run 1500 anymal_b && run 1500 a1
cat "$OUT"
