#!/bin/bash
# bash tools/ab_sweep.sh <outdir> <envs,comma> <lib> [<lib> ...]: tools/bench_sweep.py on the locomotion task for each A/B library of tools/diag
# (liblm_engine_<lib>.so; "product" = the in-tree library) -> <outdir>/sweep_<lib>.jsonl
set -e -o pipefail
OUT=$1; ENVS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $OUT
for lib in "$@"; do
  if [ $lib = product ]; then unset LM_ENGINE_SO; else export LM_ENGINE_SO=$ROOT/tools/diag/liblm_engine_$lib.so; fi
  timeout -k 10 150 python $ROOT/tools/bench_sweep.py $ENVS > $OUT/sweep_$lib.jsonl 2> $OUT/sweep_$lib.err
  python - $OUT/sweep_$lib.jsonl $lib <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print(f"{sys.argv[2]:10s}", "  ".join(f"{r['envs']}: {r['us_per_step']:.1f} us ({r['M_env_steps_per_s']:.0f} M)" for r in rows), flush=True)
PY
done
