#!/bin/bash
# round-2 experiment run 7: K-split slice count rounded down (one round of workgroups), 128x256 3-stage tile candidate
set -u
OUT=gpurun_out/r02_run7; mkdir -p $OUT
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet.py -m gpu -x -q -k "not sdxl and not real" 2>&1 | tail -8 | tee $OUT/pytest_kernels.txt
prof() { TUNE_MASK=$3 TAIL_SPLIT=$4 timeout 300 python scripts/profile_unet.py $1 $2 > $OUT/prof_$1_$2_$5.txt 2>&1; grep "^# " $OUT/prof_$1_$2_$5.txt; }
echo "== sd15 rows 16: no-15+ceil / no-15+floor / all+floor"
prof sd15 16 0xffff7fff 2 a
prof sd15 16 0xffff7fff 1 b
prof sd15 16 0xffffffff 1 c
echo "== sdxl rows 4: no-15+floor / all+floor"
prof sdxl 4 0xffff7fff 1 b
prof sdxl 4 0xffffffff 1 c
echo "== sdxl rows 2: no-15+ceil / all+floor"
prof sdxl 2 0xffff7fff 2 a
prof sdxl 2 0xffffffff 1 c
du -sh $OUT
