#!/bin/bash
# round 3, call z: instruction counters of the few-chains kernels (one and two proposals per step), 64 chains
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_few
mkdir -p $O
P="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
for pps in 1 2; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/pps$pps -o p -- python3 $R/tools/few_chains_pmc.py $pps > $O/pps$pps.log 2>&1 || { echo "pps$pps FAILED"; tail -5 $O/pps$pps.log; exit 1; }
  grep pps $O/pps$pps.log | tail -1
done
find $O -name "*agent_info.csv" -delete
python3 - <<PY
import csv, collections, glob
for pps in (1, 2):
    f = glob.glob("$O/pps%d/**/p_counter_collection.csv" % pps, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if "free_stage" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    # the two largest dispatches = the 100-sweep launches of the sample stage
    print("pps", pps, {c: sorted(v.values())[-1] for c, v in acc.items()}, "kernels", set(r for r in []))
PY
