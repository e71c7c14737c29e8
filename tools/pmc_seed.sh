#!/bin/bash
# PMC passes over the seed stage of the default workload (tools/seed_probe.py), counters only.
# usage (on the GPU box, from the repo root): bash tools/pmc_seed.sh TAG
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/seed_probe.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
    echo "pass $i done"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen:
        cnt[(k, f)] += 1
with open("$OUT/summary.txt", "w") as out:
    for k, v in agg.items():
        if k.startswith("seed_search") or k.startswith("vote") or k.startswith("decide") or k.startswith("pack2bit"):
            out.write(k + "\n")
            for c, x in sorted(v.items()):
                out.write("   %-28s %.4g\n" % (c, x))
print(open("$OUT/summary.txt").read())
PY
