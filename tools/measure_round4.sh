#!/bin/bash
# Documentation numbers of round 4 (not the headline bench):
#   bash tools/measure_round4.sh   -> gpurun_out/measure_r04/*
# throughput vs env count / task family, phase costs, rollouts with the policy in the loop (BASELINE configs 2-5 and the PD-actuator tasks), the PPO
# acceptance runs of the task families whose kernel changed this round (k_step_pd: logged torque clipped), the plate / body intersection census
#  and the GNN behaviour-cloning probe.
set -o pipefail
OUT=gpurun_out/measure_r04; mkdir -p $OUT
python tools/phase_cost.py > $OUT/phase_cost.json 2> $OUT/phase.err && echo phase ok
python tools/bench_sweep.py > $OUT/sweep.jsonl 2> $OUT/sweep.err && echo sweep ok
python tools/bench_rollout.py --policy mlp > $OUT/rollout_mlp.json 2> $OUT/rollout_mlp.err && echo mlp ok
python tools/bench_rollout.py --policy gnn > $OUT/rollout_gnn.json 2> $OUT/rollout_gnn.err && echo gnn ok
python tools/bench_rollout.py --task JointLocomanipulation --policy mlp > $OUT/rollout_config4_cotrain_mlp_4096.json 2> $OUT/c4.err && echo c4 ok
python tools/bench_rollout.py --task JointLocomanipulationVertical --num-envs 8192 --policy gnn > $OUT/rollout_config5_vertical_gnn_8192.json 2> $OUT/c5.err && echo c5 ok
python tools/bench_rollout.py --task QuadrupedManipulatePlate --policy mlp > $OUT/rollout_config3_mani_mlp_4096.json 2> $OUT/c3.err && echo c3 ok
for t in QuadrupedPoseControlCustomController QuadrupedManipulatePlateCustomController JointLocomanipulationPositionControl; do
  python tools/bench_rollout.py --task $t --policy mlp > $OUT/rollout_pd_$t.json 2> $OUT/pd_$t.err && echo "$t ok"
done
for t in QuadrupedPoseControl QuadrupedManipulatePlate JointLocomanipulation QuadrupedPoseControlCustomController QuadrupedManipulatePlateCustomController JointLocomanipulationPositionControl; do
  python tools/train_ppo.py --task $t --timesteps 9600 --log-every 50 --out $OUT/ppo_$t.json > $OUT/ppo_$t.log 2>&1 && echo "ppo $t: $(tail -1 $OUT/ppo_$t.log | cut -c1-160)"
done
python tools/plate_intersection.py --timesteps 9600 --out $OUT/plate_intersection.json > $OUT/plate.log 2>&1 && echo plate ok
python tools/gnn_bc_probe.py --out $OUT/gnn_bc_probe.json > $OUT/bc.log 2>&1 && echo bc ok; tail -3 $OUT/bc.log | cut -c1-400
tail -3 $OUT/sweep.jsonl
