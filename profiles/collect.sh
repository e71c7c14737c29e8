#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: bench lines, rocprofv3 kernel-trace stats of the
# same commands, and the PMC passes (own runs, counters only -- never combined with trace domains).
# Outputs land in gpurun_out/$TAG/ ; profiles/summarize_pmc.py turns them into profiles/<round>/.
#   default command  = steps alternate over 3 HIP streams (kernels of different steps overlap), then a serialized
#                      replay on one stream (what `roofline` is taken from) and the PCIe-inclusive leg
#   --streams 1      = one stream: every kernel has the chip to itself (isolated per-kernel durations)
# usage: bash profiles/collect.sh TAG PART      PART = A (bench lines + traces), B (PMC passes), C (other configs),
#                                               D (PMC passes of seed_search / vote on the GRCh38-sized text)
set -o pipefail
TAG=${1:-r3}
PART=${2:-A}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$PART" = "A" ]; then
    timeout -k 10 700 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
    grep -v "lrm build" $OUT/bench_default.err | tail -4
    timeout -k 10 300 python bench.py --streams 1 --cpu-seconds 5 --no-pcie --no-grch38 > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err || exit 1
    cd /tmp && export TMPDIR=/tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
        python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-pcie --no-grch38 > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_streams1 -- \
        python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --cpu-seconds 0 --no-pcie --no-grch38 --no-isolated-replay > $OUT/bench_trace_streams1.json 2> $OUT/bench_trace_streams1.err || exit 1
    echo "traces done"
elif [ "$PART" = "B" ]; then
    cd /tmp && export TMPDIR=/tmp
    for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
                "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
                "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT"; do
        tag=$(echo $pass | cut -d" " -f1)
        timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$tag -- \
            python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 1 --warmup 2 --cpu-seconds 0 --no-kernel-timing --no-pcie --no-grch38 --no-isolated-replay \
            > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err || exit 1
        echo "pmc $tag done"
    done
elif [ "$PART" = "D" ]; then
    cd /tmp && export TMPDIR=/tmp
    for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        tag=$(echo $pass | cut -d" " -f1)
        timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $OUT/grch38_pmc_$tag -- \
            python3 $GRAFT_REPO_ROOT/bench.py --ref-len 3099750718 --streams 1 --steps 1 --warmup 2 --cpu-seconds 0 --no-kernel-timing --no-pcie --no-grch38 --no-isolated-replay \
            > $OUT/grch38_pmc_$tag.json 2> $OUT/grch38_pmc_$tag.err || exit 1
        echo "grch38 pmc $tag done"
    done
else
    timeout -k 10 400 python bench.py --no-grch38 --ref-len 248956422 --reads 50000 --read-len 15000 --profile pacbio --steps 6 --warmup 2 \
        --cpu-seconds 5 > $OUT/bench_chr1_pacbio15k.json 2> $OUT/bench_chr1_pacbio15k.err || exit 1
    tail -1 $OUT/bench_chr1_pacbio15k.err
    timeout -k 10 400 python bench.py --no-grch38 --reads 20000 --read-len 100000 --steps 6 --warmup 2 --cpu-seconds 5 \
        > $OUT/bench_ultralong_20k_x_100kbp.json 2> $OUT/bench_ultralong_20k_x_100kbp.err || exit 1
    tail -1 $OUT/bench_ultralong_20k_x_100kbp.err
fi
