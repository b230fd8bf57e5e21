#!/bin/bash
# rocprofv3 kernel stats of the three rollout modes with the MLP policy in the loop (graph, persistent) -> gpurun_out/prof_rollout_<tag>/
set -e -o pipefail
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_rollout_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/tools/bench_rollout.py --policy mlp --reps 10 > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv | cut -c1-160
