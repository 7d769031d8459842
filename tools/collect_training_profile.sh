#!/bin/bash
# rocprofv3 evidence for the training loop (rollout + PPO update) on the GPU box, run through gpurun from the repo root:
#   1. --kernel-trace --stats of tools/train_probe.py (100 iterations of anymal_c_flat, 4096 envs)
#   2. tools/mlp_probe.py: learner-kernel timings against torch autograd + the in-kernel phase trace
#   3. one --pmc pass over the learner kernels (MFMA busy cycles, LDS activity)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "[1/3] kernel trace of the training loop"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/train_probe.py" 100 > "$OUT/train.log" 2> "$OUT/train.err" || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/training_kernel_stats.csv"
rm -rf "$OUT/stats"
echo "[2/3] learner kernels vs autograd"
TRACE=1 timeout -k 10 120 python3 "$ROOT/tools/mlp_probe.py" > "$OUT/mlp_probe.txt" 2> "$OUT/mlp_probe.err" || exit 1
echo "[3/3] pmc"
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/tools/mlp_probe.py" > "$OUT/pmc.log" 2> "$OUT/pmc.err" || exit 1
python3 "$ROOT/tools/pmc_kernel_means.py" "$OUT/pmc" > "$OUT/learner_pmc.txt"
rm -rf "$OUT/pmc"
tail -3 "$OUT/train.log"; cat "$OUT/mlp_probe.txt" "$OUT/learner_pmc.txt"
