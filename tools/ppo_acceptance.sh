set -o pipefail
mkdir -p gpurun_out/r2ppo
for spec in "QuadrupedPoseControl mlp 9600 loco_mlp" "QuadrupedManipulatePlate mlp 9600 mani_mlp" "JointLocomanipulation mlp 24000 cotrain_mlp" "JointLocomanipulationVertical mlp 24000 cotrain_vertical_mlp" "QuadrupedPoseControlVertical mlp 19200 loco_vertical_mlp"; do
  set -- $spec
  timeout -k 10 300 python tools/train_ppo.py --task $1 --policy $2 --timesteps $3 --num-envs 4096 --log-every 25 --out gpurun_out/r2ppo/r02_ppo_$4.json > gpurun_out/r2ppo/$4.log 2>&1 || echo "FAILED $4"
  tail -1 gpurun_out/r2ppo/$4.log | cut -c1-300
done
