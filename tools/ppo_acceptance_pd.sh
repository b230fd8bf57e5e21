# Behavioural acceptance of the PD-actuator families after the clamp-decision change: the reference's PPO recipe on the five f-1 tasks.
#   bash tools/ppo_acceptance_pd.sh   (GPU box)  -> gpurun_out/r3ppo_pd/r03_ppo_*.json
set -o pipefail
mkdir -p gpurun_out/r3ppo_pd
for spec in "QuadrupedPoseControlCustomController mlp 9600 loco_cc_mlp" "QuadrupedManipulatePlateCustomController mlp 9600 mani_cc_mlp" "JointLocomanipulationPositionControl mlp 9600 cotrain_pc_mlp" \
            "QuadrupedPoseControlPositionControl mlp 9600 loco_pc_mlp" "QuadrupedManipulatePlatePositionControl mlp 9600 mani_pc_mlp"; do
  set -- $spec
  timeout -k 10 400 python tools/train_ppo.py --task $1 --policy $2 --timesteps $3 --num-envs 4096 --log-every 25 --out gpurun_out/r3ppo_pd/r03_ppo_$4.json > gpurun_out/r3ppo_pd/$4.log 2>&1 || echo "FAILED $4"
  echo "$4: $(tail -1 gpurun_out/r3ppo_pd/$4.log | cut -c1-220)"
done
