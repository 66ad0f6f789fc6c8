#!/bin/bash
# round-2 experiment run 8: 8-wave 128x128 tile (two workgroups / CU) + 256x64 as tuner candidates; tile of the rule-based K-split launches
set -u
OUT=gpurun_out/r02_run8; mkdir -p $OUT
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "big or geglu or heads or tail or tuning or all_tile" 2>&1 | tail -6 | tee $OUT/pytest_kernels.txt
prof() { TUNE_MASK=$3 SPLIT_TILE=$4 timeout 300 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_$5.txt 2>&1; grep "^# " $OUT/prof_$1_$2_$5.txt; }
echo "== sd15 rows 16: no-17/2 | all | all+split14 | all+split12"
prof sd15 16 0xfffdfffb 1 a
prof sd15 16 0xffffffff 1 b
prof sd15 16 0xffffffff 14 c
prof sd15 16 0xffffffff 12 d
echo "== sdxl rows 4: no-17/2 | all"
prof sdxl 4 0xfffdfffb 1 a
prof sdxl 4 0xffffffff 1 b
du -sh $OUT
