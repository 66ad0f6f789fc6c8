#!/bin/bash
# Round-5 call 14: HBM traffic per implicit-GEMM launch class (FETCH / WRITE passes over UNet-only forwards, folded by position onto
# the plan's launches: scripts/pmc_per_class.py) for the two default-bench populations
set -u
OUT=gpurun_out/r05_call14; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
R=$GRAFT_REPO_ROOT
for spec in "sd15 16" "sdxl 4"; do
  set -- $spec; cfg=$1; rows=$2
  timeout 400 python scripts/pmc_unet.py $cfg $rows --save-hints > $OUT/${cfg}_hints.log 2>&1
  cd /tmp
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/${cfg}_fetch -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/${cfg}_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/${cfg}_write -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/${cfg}_write.log 2>&1
  cd $R
  python scripts/pmc_per_class.py --fetch $OUT/${cfg}_fetch --write $OUT/${cfg}_write --detail gpurun_out/detail_${cfg}_rows${rows}.txt --rows $rows > $OUT/pmc_per_launch_class_${cfg}_rows${rows}.txt 2>&1
  head -32 $OUT/pmc_per_launch_class_${cfg}_rows${rows}.txt | cut -c1-150
  find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
done
