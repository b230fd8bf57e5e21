set -o pipefail
for spec in "QuadrupedPoseControl 9600" "QuadrupedManipulatePlate 9600" "JointLocomanipulation 14400" "JointLocomanipulationVertical 24000"; do
  set -- $spec
  timeout -k 10 200 python tools/train_ppo.py --task $1 --timesteps $2 --num-envs 4096 --log-every 1000 --seed 42 2>/dev/null | grep iteration | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'task': '$1', 'timesteps': d['timesteps'], 'success_rate': round(d['success_rate'], 4), 'mean_reward': round(d['mean_reward'], 3), 'wall_s': round(d['wall_s'], 1)}))"
done
timeout -k 10 300 python tools/train_ppo.py --task JointLocomanipulationVertical --policy gnn --timesteps 2400 --num-envs 4096 --log-every 600 --seed 42 2>/dev/null | grep iteration | tail -1 | cut -c1-300
