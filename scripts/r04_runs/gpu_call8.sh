#!/bin/bash
# Round-4 call 8: the Lightning (SDXL 16 rows) and edit (SDXL 2 rows) populations with / without the round-4 tile candidates, same box
set -u
OUT=gpurun_out/r04_call8; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
OLD=0xffec5fff
timeout 1200 python scripts/r04_runs/ab_forward.py sdxl 8 "none:mask=$OLD;t32_all:mask=0xffffffff" > $OUT/ab_sdxl_b8.txt 2>&1; cat $OUT/ab_sdxl_b8.txt | cut -c1-420
timeout 900 python scripts/r04_runs/ab_forward.py sdxl 1 "none:mask=$OLD;t32_all:mask=0xffffffff" > $OUT/ab_sdxl_b1.txt 2>&1; cat $OUT/ab_sdxl_b1.txt | cut -c1-420
