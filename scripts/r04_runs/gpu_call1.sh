#!/bin/bash
# Round-4 call 1: first contact of the merged inline-LayerNorm-statistics kernels at model level, the lane engine, the packed
# GELU; A/B of whole forwards (same box, same process); VALU issue-rate micro-benchmark.
set -u
OUT=gpurun_out/r04_call1; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
echo "== micro VALU"; timeout 120 scripts/r04_runs/micro_valu.bin > $OUT/micro_valu.txt 2>&1; cat $OUT/micro_valu.txt
echo "== kernel + lane + semantics tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lanes.py tests/test_gpu_torch_semantics.py -x -q -m gpu > $OUT/pytest_a.txt 2>&1; tail -5 $OUT/pytest_a.txt
echo "== model-level parity with LayerNorm statistics in the K loop (CFGPP_FUSE_LN=2)"
CFGPP_FUSE_LN=2 timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "forward_vs_oracle or broadcast or sd_chain" > $OUT/pytest_ln2.txt 2>&1; tail -5 $OUT/pytest_ln2.txt
echo "== A/B sd15 b8"; timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "0:1,2:1,0:2,2:2,2:4" > $OUT/ab_sd15_b8.txt 2>&1; cat $OUT/ab_sd15_b8.txt
echo "== A/B sdxl b2"; timeout 1200 python scripts/r04_runs/ab_forward.py sdxl 2 "0:1,2:1,2:2,2:4" > $OUT/ab_sdxl_b2.txt 2>&1; cat $OUT/ab_sdxl_b2.txt
echo "== per-launch table sd15 rows16, fuse_ln 2"; FUSE_LN=2 timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/unet_launch_profile_sd15_rows16_ln2.txt 2>&1; head -40 $OUT/unet_launch_profile_sd15_rows16_ln2.txt
