#!/bin/bash
# PMC passes for k_gnn_forward at 8192 samples (tools/bench_gnn.py); one counter group per run, kernel trace only.
set -e -o pipefail
TAG=${1:-a}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_gnn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o run -- python3 $ROOT/tools/bench_gnn.py > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(list); grid = 0
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_gnn_forward"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
waves = grid // 64
print(json.dumps({"kernel": "k_gnn_forward", "samples": 8192, "wavefronts": waves, "per_wave": {k: round(sum(v) / len(v) / waves, 1) for k, v in sorted(acc.items())}}, indent=1))
PY
