#!/bin/bash
# round-2 GPU run 5: staged heads epilogue, two-K-tiles-per-barrier tile, raw-barrier 3-stage ring; quick validation
set -u
OUT=gpurun_out/r02_run5; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu (quick subset)"; timeout 1500 python -m pytest tests -m gpu -q -k "not real_" 2>&1 | tail -30 > $OUT/pytest_gpu_quick.txt; tail -8 $OUT/pytest_gpu_quick.txt
echo "== heads epilogue A/B (STAGED=0 plain / 1 LDS-staged)"
for st in 0 1; do for s in "heads_m4096 4" "heads_m4096 7" "heads_m4096 1" "heads_l0 6" "heads_l0 8" "heads_l1 8" "heads_l1 4"; do STAGED=$st timeout 60 python scripts/one_igemm.py $s 2>&1 | tail -1 | sed "s/^/staged=$st /" | tee -a $OUT/heads_ab.txt; done; done
echo "== M=4096 tiles: 7 (2-stage), 9 (3-stage, raw barrier), 11 (two tiles per barrier)"
for s in "lin_m4096_1280" "lin_m4096_ffout" "conv_l2_1280" "conv_l1_640"; do for c in 7 9 11 1; do timeout 60 python scripts/one_igemm.py $s $c 2>&1 | tail -1 | tee -a $OUT/tiles_ab.txt; done; done
echo "== profile sd15"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15.txt 2>&1; head -3 $OUT/prof_sd15.txt; tail -1 $OUT/prof_sd15.txt
echo "== profile sdxl"; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl.txt 2>&1; head -3 $OUT/prof_sdxl.txt; tail -1 $OUT/prof_sdxl.txt
echo "== sd15 rows=16 forward vs oracle"; timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "real_unet_forward_at_bench_size and sd15" 2>&1 | tail -3
du -sh $OUT
