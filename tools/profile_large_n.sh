#!/bin/bash
# rocprofv3 kernel statistics of the large-N step kernel (k_step_w2, beyond 32 768 locomotion envs) and of k_step at the same sizes:
#   bash tools/profile_large_n.sh   -> gpurun_out/prof_large_n/{w2,w1}_kernel_stats.csv
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_large_n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w2 -o run -- python3 $ROOT/tools/bench_sweep.py 131072 > $OUT/w2.log 2>&1
export LM_W2_MIN_ENVS=1000000000
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w1 -o run -- python3 $ROOT/tools/bench_sweep.py 131072 > $OUT/w1.log 2>&1
for v in w2 w1; do find $OUT/$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${v}_kernel_stats.csv; head -3 $OUT/${v}_kernel_stats.csv | cut -c1-120; done
