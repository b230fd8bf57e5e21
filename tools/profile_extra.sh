#!/bin/bash
# Extra PMC passes for k_step (instruction cache, instruction-issue stalls, instruction classes); same protocol as tools/profile_round.sh.
set -e -o pipefail
TAG=${1:-x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_extra_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/g$i -o run -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --timed-only > $OUT/g$i.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4 $OUT/g5
