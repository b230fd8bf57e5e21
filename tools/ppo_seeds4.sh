# seed robustness of the behavioural acceptance runs under the round-4 build (policy tiles on the fp16 matrix pipe, k_step_pd with the clipped logged torque): 3 more seeds per task, final success rate
set -o pipefail
mkdir -p gpurun_out/r4ppo
for spec in "QuadrupedPoseControl 9600" "QuadrupedManipulatePlate 9600" "JointLocomanipulation 14400" "JointLocomanipulationVertical 24000" "QuadrupedManipulatePlateCustomController 9600" "JointLocomanipulationPositionControl 9600" "QuadrupedPoseControlCustomController 24000"; do
  set -- $spec
  for seed in 1 2 3; do
    timeout -k 10 200 python tools/train_ppo.py --task $1 --timesteps $2 --num-envs 4096 --log-every 1000 --seed $seed 2>/dev/null | grep iteration | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'task': '$1', 'seed': $seed, 'timesteps': d['timesteps'], 'success_rate': round(d['success_rate'], 4), 'mean_reward': round(d['mean_reward'], 3), 'wall_s': round(d['wall_s'], 1)}))"
  done
done
