#!/bin/bash
# bash tools/w2_probe.sh <envs>: wavefront lifetime / VALU work / HBM traffic of k_step at <envs> for the product library and the two-wavefronts-per-SIMD
# A/B build (tools/ab_build.py w2=-DLM_WAVES2), rocprofv3 --pmc in separate passes (never combined with other trace domains) -> gpurun_out/w2_probe/
set -e -o pipefail
N=${1:-32768}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/w2_probe; mkdir -p $OUT
for lib in product w2; do
  if [ $lib = w2 ]; then export LM_ENGINE_SO=$ROOT/tools/diag/liblm_engine_w2.so; else unset LM_ENGINE_SO; fi
  bash $ROOT/tools/diag/pmc_kernel.sh k_step tools/bench_sweep.py $N > $OUT/pmc_${lib}_$N.json
  rm -rf $OUT/raw_$lib; mv $ROOT/gpurun_out/pmc_k_step $OUT/raw_$lib
  cat $OUT/pmc_${lib}_$N.json
done
