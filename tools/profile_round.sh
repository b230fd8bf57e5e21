#!/bin/bash
# Profiles of the headline bench for profiles/ (run on the GPU box through gpurun):
#   bash tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{stats,pmc_*}/...
# kernel-trace/stats and each --pmc group are separate rocprofv3 runs (counters never combined with other trace domains).
set -e -o pipefail
TAG=${1:-v4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline --timed-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32" "SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/pmc_$name -o run -- python3 $ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --timed-only > $OUT/pmc_$name.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py --json $OUT > $OUT/pmc_summary.json
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
ls $OUT
