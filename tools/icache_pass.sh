# Instruction-cache counters of the step kernel (one --pmc pass, eager launches); summary kept as profiles/r02_pmc_icache.txt
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/icache
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*" | sort -u > $OUT/avail_icache.txt
timeout -k 10 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_icache -- python3 $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --training-iters 0 --no-graph --event-steps 20 > $OUT/pmc_icache.json 2> $OUT/pmc_icache.err
echo rc=$?
