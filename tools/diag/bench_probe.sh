set -e
P='import sys,json; r=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("%s value %.1f M  event %.1f M"%(sys.argv[1], r["value"]/1e6, r["config"]["event_timed_env_steps_per_s_rank0"]/1e6))'
for i in 1 2 3 4; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "$P" plain; done
for i in 1 2 3 4; do LM_BENCH_SPIN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "$P" spin; done
for i in 1 2 3 4; do GPU_MAX_HW_QUEUES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --timed-only 2>/dev/null | python -c "$P" hwq1; done
