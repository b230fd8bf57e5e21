# nine more seeds (4..12) of the six task families that learned on seeds 1-3 (tools/ppo_seeds4.sh): is "every seed" still true at twelve?
set -o pipefail
for spec in "QuadrupedPoseControl 9600" "QuadrupedManipulatePlate 9600" "JointLocomanipulation 14400" "JointLocomanipulationVertical 24000" "QuadrupedManipulatePlateCustomController 9600" "JointLocomanipulationPositionControl 9600"; do
  set -- $spec
  for seed in 4 5 6 7 8 9 10 11 12; do
    timeout -k 10 200 python tools/train_ppo.py --task $1 --timesteps $2 --num-envs 4096 --log-every 1000 --seed $seed 2>/dev/null | grep iteration | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'task': '$1', 'seed': $seed, 'timesteps': d['timesteps'], 'success_rate': round(d['success_rate'], 4), 'mean_reward': round(d['mean_reward'], 3), 'wall_s': round(d['wall_s'], 1)}))"
  done
done
