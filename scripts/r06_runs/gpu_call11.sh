#!/bin/bash
# Round-6 call 11 (shipped build): PMC passes again with the persistent tile counted in the implicit-GEMM family
# (scripts/pmc_summary.py matched "big4_kernel", which "big4p_kernel" does not contain: call 9's summaries missed its launches),
# then the default bench line with those summaries in its roofline block
set -u
python -c "from cfgpp_amd import _lib; print(_lib.build_id())"
bash scripts/r06_runs/gpu_final.sh pmc
OUT=gpurun_out/r06_final
echo "== bench (default)"; timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-400
