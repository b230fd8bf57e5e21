#!/bin/bash
# What do the k_step wavefronts of one CU share?  PMC groups (separate rocprofv3 passes) at 4096 envs (one wavefront per CU) and 16384 (four per CU).
#   bash tools/diag/pmc_sharing.sh  -> gpurun_out/pmc_sharing/summary.json
set -o pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/pmc_sharing
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for N in 4096 16384; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_IFETCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU"; do
    i=$((i+1)); rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/n$N/g$i -o run -- python3 $ROOT/tools/bench_sweep.py $N > $OUT/n${N}_g$i.log 2>&1 || echo "N $N group $i ($grp) failed"
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = {}
for N in (4096, 16384):
    acc = collections.defaultdict(list); grid = 0
    for f in glob.glob(f"{sys.argv[1]}/n{N}/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_step"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); grid = int(r["Grid_Size"])
    waves = max(grid // 64, 1)
    out[N] = {"wavefronts": waves, "per_wave": {k: round(sum(v) / len(v) / waves, 1) for k, v in sorted(acc.items())}}
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1); print(json.dumps(out))
PY
