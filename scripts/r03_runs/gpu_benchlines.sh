#!/bin/bash
# Round 3: the other two BASELINE workloads on the shipped build (the default command already carries SD1.5 b8 + the SDXL b2 leg)
set -u
OUT=gpurun_out/r03_benchlines; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
for c in sdxl_lightning sdxl_edit; do
  timeout 420 python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_$c.json"))
    print("$c", r["value"], r["unit"], r["ms_per_step"], "ms/job; igemm", r["roofline"]["achieved"], "TF/s; attention", r["roofline"].get("attention_TFLOPs"))
except Exception as e:
    print("$c failed:", e)
PY
done
