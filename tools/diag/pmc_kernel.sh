#!/bin/bash
# bash tools/diag/pmc_kernel.sh <kernel-name-prefix> <python script> [args]: PMC groups for one kernel, per wavefront
set -e -o pipefail
K=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$K
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o run -- python3 $ROOT/"$@" > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - $OUT $K <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(list); grid = 0
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
waves = max(grid // 64, 1)
print(json.dumps({"kernel": sys.argv[2], "wavefronts": waves, "per_wave": {k: round(sum(v) / len(v) / waves, 1) for k, v in sorted(acc.items())}}))
PY
