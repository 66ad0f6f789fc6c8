#!/bin/bash
# Round 3, GPU call 6: (a) time-embedding segments sized at run time (feature maps under 8 x 8: the tiny test models failed in
# call 5), (b) the rewritten LayerNorm / LN-stats kernels, (c) A/B: parameter DMAs before / after the prologue's tiles,
# LayerNorm fusion off / on - kernel tests, model parity with each variant, timelines, in-situ profiles.
set -u
OUT=gpurun_out/r03_call6; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 tests"
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_kernels.txt
timeout 500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_unet_vae.txt
CFGPP_FUSE_LN=1 CFGPP_PAR_LATE=1 timeout 500 python -m pytest tests/test_gpu_unet.py -m gpu -q -k "forward_vs_oracle or deterministic or chain" 2>&1 | tail -12 | tee $OUT/pytest_unet_fuse_ln_par_late.txt
echo "== 2 timelines"
for late in 0 1; do
  PAR_LATE=$late timeout 300 python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" "heads HW=1024 N=3840 K=1280" \
      "conv3x3 amode=1 HW=1024 N=1280 K=11520 +temb" > $OUT/timeline_sdxl_rows4_late$late.txt 2>&1
  echo "-- par_late=$late"; grep -v amdgpu.ids $OUT/timeline_sdxl_rows4_late$late.txt | grep -E "^##|issuing|prologue \(|per K-tile|epilogue|workgroup total|span"
done
echo "== 3 in-situ"
prof() { name=$1; shift; env "$@" timeout 200 python scripts/profile_unet.py ${CFG} > $OUT/prof_${CFGN}_$name.txt 2>&1; echo "$name: $(grep '^# ' $OUT/prof_${CFGN}_$name.txt | head -2 | tr '\n' ' ')"; }
for c in "sd15 16" "sdxl 4"; do
  CFG="$c"; CFGN=$(echo $c | tr ' ' '_')
  echo "-- $c"
  prof base FUSE_LN=0 PAR_LATE=0
  prof late FUSE_LN=0 PAR_LATE=1
  prof ln FUSE_LN=1 PAR_LATE=0
  prof ln_late FUSE_LN=1 PAR_LATE=1
done
du -sh $OUT
