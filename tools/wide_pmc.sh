# PMC passes over the wide learner kernels (tools/wide_probe.py 235: one forward + backward of actor + critic, 24 576 rows):
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as the hardware requires) and issue counters.  Summary: profiles/r02_wide_mlp_pmc.txt
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/wide_pmc
mkdir -p $OUT
export PYTHONPATH=$ROOT WIDE_ITERS=6
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    echo "[$i/3] $grp"
    timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/wide_probe.py 235 > $OUT/p$i.txt 2> $OUT/p$i.err || exit 1
    python3 $ROOT/tools/pmc_by_grid.py $OUT/p$i "lg::" > $OUT/p$i.summary.txt
done
