set -o pipefail
mkdir -p gpurun_out/r2ppo
for spec in "QuadrupedPoseControlCustomController mlp 9600 loco_cc_mlp" "QuadrupedManipulatePlateCustomController mlp 9600 mani_cc_mlp" "JointLocomanipulationPositionControl mlp 9600 cotrain_pc_mlp" "QuadrupedManipulatePlateVertical mlp 19200 mani_vertical_mlp" "JointLocomanipulation mlp 72000 cotrain_mlp_72k" "JointLocomanipulationVertical mlp 72000 cotrain_vertical_mlp_72k"; do
  set -- $spec
  timeout -k 10 400 python tools/train_ppo.py --task $1 --policy $2 --timesteps $3 --num-envs 4096 --log-every 25 --out gpurun_out/r2ppo/r02_ppo_$4.json > gpurun_out/r2ppo/$4.log 2>&1 || echo "FAILED $4"
  tail -1 gpurun_out/r2ppo/$4.log | cut -c1-200
done
