#!/bin/bash
# round-2 closing check: safetensors-file weights on the GPU + the bench line with the per-launch-minimum live profile
set -u
OUT=gpurun_out/r02_mini; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_weights.py -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest_weights.txt
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_sd15_b8_nocpu.json 2> $OUT/bench_sd15_b8_nocpu.err; cat $OUT/bench_sd15_b8_nocpu.json | cut -c1-1500; grep "profiled passes" $OUT/bench_sd15_b8_nocpu.err
