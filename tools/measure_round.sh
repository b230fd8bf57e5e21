#!/bin/bash
# Documentation numbers of a round (not the headline bench): throughput vs env count / task family, rollouts with the policy in the loop.
#   bash tools/measure_round.sh <tag>   -> gpurun_out/measure_<tag>/*.json
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/measure_$TAG; mkdir -p $OUT
python tools/bench_sweep.py > $OUT/sweep.jsonl 2> $OUT/sweep.err && echo sweep ok
python tools/bench_rollout.py --policy mlp > $OUT/rollout_mlp.json 2> $OUT/rollout_mlp.err && echo mlp ok
python tools/bench_rollout.py --policy gnn > $OUT/rollout_gnn.json 2> $OUT/rollout_gnn.err && echo gnn ok
python tools/bench_rollout.py --task JointLocomanipulation --policy mlp > $OUT/rollout_config4_cotrain_mlp_4096.json 2> $OUT/c4.err && echo c4 ok
python tools/bench_rollout.py --task JointLocomanipulationVertical --num-envs 8192 --policy gnn > $OUT/rollout_config5_vertical_gnn_8192.json 2> $OUT/c5.err && echo c5 ok
python tools/bench_rollout.py --task QuadrupedManipulatePlate --policy mlp > $OUT/rollout_config3_mani_mlp_4096.json 2> $OUT/c3.err && echo c3 ok
python tools/phase_cost.py > $OUT/phase_cost.log 2>&1 && echo phase ok
tail -3 $OUT/sweep.jsonl
