#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel-trace stats of the
# same command, and the PMC passes (own runs, counters only -- never combined with trace domains).
# Outputs land in gpurun_out/$TAG/ ; profiles/summarize_pmc.py turns them into profiles/<round>/.
set -o pipefail
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
tail -2 $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $pass | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$tag -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-kernel-timing \
        > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || exit 1
    echo "pmc $tag done"
done
