#!/bin/bash
# bash tools/diag/cc_seeds_variant.sh <label> <first> <last> <extra train_ppo.py flags...>: the custom-controller locomotion task over seeds, with a trainer-side variation
LABEL=$1; A=$2; B=$3; shift 3
for seed in $(seq $A $B); do
  timeout -k 10 200 python tools/train_ppo.py --task QuadrupedPoseControlCustomController --timesteps 24000 --num-envs 4096 --log-every 100 --seed $seed "$@" 2>/dev/null | grep iteration | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin]
first=next((r['timesteps'] for r in rows if r['success_rate']>=0.95), None)
d=rows[-1]; print(json.dumps({'variant': '$LABEL', 'seed': $seed, 'success_rate': round(d['success_rate'], 4), 'first_0.95_at': first, 'mean_reward': round(d['mean_reward'], 3), 'std': round(d['std'], 3)}))"
done
