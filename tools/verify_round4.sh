#!/bin/bash
# What the driver runs at round end, plus the policy-tile numbers, on the current in-tree build:  bash tools/verify_round4.sh -> gpurun_out/r4h/
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/microbench/stream_copy.bin > $OUT/stream_copy.json; cat $OUT/stream_copy.json
python bench.py > $OUT/bench_final.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_protocol.json 2>/dev/null
python tools/bench_gnn.py 2>/dev/null > $OUT/gnn_bench.json
python tools/stamp_profile_gnn.py 2>/dev/null > $OUT/gnn_stamps.json
python tools/bench_rollout.py --task JointLocomanipulationVertical --num-envs 8192 --policy gnn 2>/dev/null > $OUT/rollout_config5.json
python tools/bench_rollout.py --policy gnn 2>/dev/null > $OUT/rollout_gnn.json
python - <<'PY'
import json
for f in ("bench_final", "bench_driver_protocol"):
    r = json.load(open("gpurun_out/r4h/%s.json" % f)); print(f, round(r["value"] / 1e6, 1), {k: round(v / 1e6, 1) for k, v in r["config"].items() if isinstance(v, float) and v > 1e6})
for f in ("rollout_config5", "rollout_gnn"):
    d = json.load(open("gpurun_out/r4h/%s.json" % f)); print(f, {k: round(v["us_per_step"], 1) for k, v in d.items() if isinstance(v, dict) and "us_per_step" in v})
print(open("gpurun_out/r4h/gnn_bench.json").read()[:120])
PY
