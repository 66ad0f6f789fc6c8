#!/bin/bash
# Round-5 call 3: does the 128 x 320 one-wave-per-SIMD tile (config 25) beat the 16x16x32 rule on the two-round class
# (M = 16384 x N = 640) in situ?  rounds=1 hands that class to the tuner (incl. 24 - 26); mf16=0 hands it everything.
set -u
OUT=gpurun_out/r05_call3; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
V="base:mask=0xf8ffffff;r1_big4:rounds=1;r1_nobig4:rounds=1,mask=0xf8ffffff;nomf16_big4:mf16=0"
echo "== A/B sd15 b8"; timeout 900 python scripts/r05_runs/ab_forward.py sd15 8 "$V" --table > $OUT/ab_sd15_b8.txt 2>&1; head -7 $OUT/ab_sd15_b8.txt | cut -c1-520
echo "== A/B sdxl b2"; timeout 1500 python scripts/r05_runs/ab_forward.py sdxl 2 "$V" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -7 $OUT/ab_sdxl_b2.txt | cut -c1-520
echo "== table sd15 (last variant)"; sed -n 7,60p $OUT/ab_sd15_b8.txt
echo "== table sdxl (last variant)"; sed -n 7,45p $OUT/ab_sdxl_b2.txt
