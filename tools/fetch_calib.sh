#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the access shapes of k_step (tools/microbench/fetch_calib.hip); separate --pmc passes.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/fetch_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/$c -o run -- $ROOT/tools/microbench/fetch_calib.bin > $OUT/$c.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]; res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): res[k][c + "_raw_per_launch"] = sum(v) / len(v)
touched = 64 * (1 << 20) * 4
print(json.dumps({"bytes_touched_per_kernel": touched, "kernels": res,
                  "ratio_raw_x1024_over_touched": {k: {c: round(v * 1024 / touched, 4) for c, v in d.items()} for k, d in res.items()}}, indent=1))
PY
