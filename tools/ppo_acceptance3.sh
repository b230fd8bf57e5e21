# Behavioural acceptance of round 3's physics (16 sweeps on the ground, paired friction rows): the reference's PPO recipe on every task family.
#   bash tools/ppo_acceptance3.sh   (GPU box)  -> gpurun_out/r3ppo/r03_ppo_*.json
set -o pipefail
mkdir -p gpurun_out/r3ppo
for spec in "QuadrupedPoseControl mlp 9600 loco_mlp" "QuadrupedManipulatePlate mlp 9600 mani_mlp" "JointLocomanipulation mlp 24000 cotrain_mlp" "JointLocomanipulationVertical mlp 24000 cotrain_vertical_mlp" \
            "QuadrupedPoseControlVertical mlp 19200 loco_vertical_mlp" "QuadrupedManipulatePlateVertical mlp 19200 mani_vertical_mlp" \
            "QuadrupedPoseControlCustomController mlp 24000 loco_cc_mlp" "QuadrupedManipulatePlateCustomController mlp 9600 mani_cc_mlp" "JointLocomanipulationPositionControl mlp 9600 cotrain_pc_mlp"; do
  set -- $spec
  timeout -k 10 400 python tools/train_ppo.py --task $1 --policy $2 --timesteps $3 --num-envs 4096 --log-every 25 --out gpurun_out/r3ppo/r03_ppo_$4.json > gpurun_out/r3ppo/$4.log 2>&1 || echo "FAILED $4"
  echo "$4: $(tail -1 gpurun_out/r3ppo/$4.log | cut -c1-220)"
done
