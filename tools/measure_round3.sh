#!/bin/bash
# Documentation numbers of round 3 (not the headline bench): throughput vs env count / task family, rollouts with the policy in the loop for the
# BASELINE configs and the five PD-actuator tasks, phase costs.
#   bash tools/measure_round3.sh   -> gpurun_out/measure_r03/*.json
set -o pipefail
OUT=gpurun_out/measure_r03; mkdir -p $OUT
python tools/phase_cost.py > $OUT/phase_cost.json 2> $OUT/phase.err && echo phase ok
python tools/bench_sweep.py > $OUT/sweep.jsonl 2> $OUT/sweep.err && echo sweep ok
python tools/bench_rollout.py --policy mlp > $OUT/rollout_mlp.json 2> $OUT/rollout_mlp.err && echo mlp ok
python tools/bench_rollout.py --policy gnn > $OUT/rollout_gnn.json 2> $OUT/rollout_gnn.err && echo gnn ok
python tools/bench_rollout.py --task JointLocomanipulation --policy mlp > $OUT/rollout_config4_cotrain_mlp_4096.json 2> $OUT/c4.err && echo c4 ok
python tools/bench_rollout.py --task JointLocomanipulationVertical --num-envs 8192 --policy gnn > $OUT/rollout_config5_vertical_gnn_8192.json 2> $OUT/c5.err && echo c5 ok
python tools/bench_rollout.py --task QuadrupedManipulatePlate --policy mlp > $OUT/rollout_config3_mani_mlp_4096.json 2> $OUT/c3.err && echo c3 ok
for t in QuadrupedPoseControlCustomController QuadrupedManipulatePlateCustomController QuadrupedPoseControlPositionControl QuadrupedManipulatePlatePositionControl JointLocomanipulationPositionControl; do
  python tools/bench_rollout.py --task $t --policy mlp > $OUT/rollout_pd_$t.json 2> $OUT/pd_$t.err && echo "$t ok"
done
tail -3 $OUT/sweep.jsonl
