#!/bin/bash
# Round 3, GPU call 8: fused-LayerNorm row statistics through LDS (the consumer epilogues' per-row global loads were
# serialised round trips).  Default path re-validated (parameter-segment layout changed), then FUSE_LN 0 / 1 in situ.
set -u
OUT=gpurun_out/r03_call8; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 tests"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_kernels.txt
timeout 400 python -m pytest tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_unet.txt
CFGPP_FUSE_LN=1 timeout 400 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "forward_vs_oracle or deterministic or chain" 2>&1 | tail -8 | tee $OUT/pytest_unet_fuse_ln.txt
echo "== 2 in-situ"
prof() { name=$1; shift; env "$@" timeout 200 python scripts/profile_unet.py ${CFG} > $OUT/prof_${CFGN}_$name.txt 2>&1; echo "$name: $(grep '^# ' $OUT/prof_${CFGN}_$name.txt | head -2 | tr '\n' ' ')"; }
for c in "sd15 16" "sdxl 4"; do
  CFG="$c"; CFGN=$(echo $c | tr ' ' '_')
  echo "-- $c"
  prof base FUSE_LN=0
  prof ln FUSE_LN=1
done
