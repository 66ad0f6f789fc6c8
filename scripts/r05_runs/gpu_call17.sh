#!/bin/bash
# Round-5 call 17: the PMC / kernel-trace passes of gpu_final.sh for the two default-bench populations, on the LAST build
# (with the XCD-blocked walk), so that profiles/r05/pmc_sd15_b8.json / pmc_sdxl_b2.json describe the shipped binary
set -u
OUT=gpurun_out/r05_final2; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
R=$GRAFT_REPO_ROOT
HEAD=$(cat $R/profiles/r05/HEAD.txt 2>/dev/null)
pmc_config() {   # unet_config rows bench_config batch
  local cfg=$1 rows=$2 bc=$3 b=$4
  timeout 300 python scripts/pmc_unet.py $cfg $rows --save-hints > $OUT/pmc_${bc}_hints.log 2>&1
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_$bc -o t --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 10 > $R/$OUT/trace_$bc.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_${bc}_fetch -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_${bc}_write -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/$OUT/pmc_${bc}_sq -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_sq.log 2>&1
  cd $R
  python scripts/pmc_summary.py --fetch $OUT/pmc_${bc}_fetch --write $OUT/pmc_${bc}_write --sq $OUT/pmc_${bc}_sq --trace $OUT/trace_$bc \
      --detail gpurun_out/detail_${cfg}_rows${rows}.txt --rows $rows --out $OUT/pmc_${bc}_b${b}.json \
      --note "round-5 shipped build ($HEAD): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) and --kernel-trace --stats over scripts/pmc_unet.py $cfg $rows --load-hints: UNet-only forwards at UNet batch $rows with the tiles the tuner pinned in the un-profiled run" > $OUT/pmc_${bc}_summary.log 2>&1
  find $OUT/trace_$bc -name "*kernel_stats.csv" -exec cp {} $OUT/${bc}_unet_only_kernel_stats.csv \;
  find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
}
echo "== PMC sdxl rows 4"; pmc_config sdxl 4 sdxl 2; grep -A14 '"igemm": {' $OUT/pmc_sdxl_b2.json | head -18
echo "== PMC sd15 rows 16"; pmc_config sd15 16 sd15 8; grep -A14 '"igemm": {' $OUT/pmc_sd15_b8.json | head -18
