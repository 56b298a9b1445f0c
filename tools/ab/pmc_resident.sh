#!/bin/bash
# usage (on the GPU box): tools/ab/pmc_resident.sh  -- SQ counters of the LDS-resident kernel (20 steps, batch 256)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_resident
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- python tools/ab/prof_resident.py > $OUT/p$i.log 2>&1
  python - <<PY
import csv, glob
agg = {}
for f in glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resident" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
print(agg)
PY
done
