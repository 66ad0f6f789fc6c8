#!/bin/bash
# the four benchmark lines on the final build (bench.py with the per-launch-minimum live profile)
set -u
OUT=gpurun_out/r02_final3; mkdir -p $OUT
timeout 100 python bench.py --config sdxl --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_sdxl_b2.json 2> $OUT/bench_sdxl_b2.err; cut -c1-120 $OUT/bench_sdxl_b2.json
timeout 100 python bench.py --config sdxl_lightning --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_sdxl_lightning_b8.json 2> $OUT/bench_sdxl_lightning_b8.err; cut -c1-120 $OUT/bench_sdxl_lightning_b8.json
timeout 100 python bench.py --config sdxl_edit --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_sdxl_edit_b1.json 2> $OUT/bench_sdxl_edit_b1.err; cut -c1-120 $OUT/bench_sdxl_edit_b1.json
timeout 100 python bench.py > $OUT/bench_sd15_b8.json 2> $OUT/bench_sd15_b8.err; cut -c1-120 $OUT/bench_sd15_b8.json
grep -h "profiled passes" $OUT/*.err
