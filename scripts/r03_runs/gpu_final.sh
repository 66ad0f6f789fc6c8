#!/bin/bash
# Round-3 final GPU run ON THE SHIPPED BUILD: PMC passes (FETCH / WRITE / SQ, each alone) + kernel trace over UNet-only forwards of
# sd15 rows 16 and sdxl rows 4 (the populations bench.py's roofline block describes), the default bench command (which also times
# the SDXL leg), rocprofv3 --kernel-trace --stats of that command, per-launch tables.   ~12 GPU-minutes.
set -u
OUT=gpurun_out/r03_final; mkdir -p $OUT profiles/r03
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
R=$GRAFT_REPO_ROOT
pmc_config() {   # name rows bench_config batch
  local cfg=$1 rows=$2 bc=$3 b=$4
  timeout 300 python scripts/pmc_unet.py $cfg $rows --save-hints > $OUT/pmc_${cfg}_hints.log 2>&1
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_$cfg -o t --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 10 > $R/$OUT/trace_$cfg.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_${cfg}_fetch -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints > $R/$OUT/pmc_${cfg}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_${cfg}_write -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints > $R/$OUT/pmc_${cfg}_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/$OUT/pmc_${cfg}_sq -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints > $R/$OUT/pmc_${cfg}_sq.log 2>&1
  cd $R
  python scripts/pmc_summary.py --fetch $OUT/pmc_${cfg}_fetch --write $OUT/pmc_${cfg}_write --sq $OUT/pmc_${cfg}_sq --trace $OUT/trace_$cfg \
      --detail gpurun_out/detail_${cfg}_rows${rows}.txt --rows $rows --out $OUT/pmc_${bc}_b${b}.json \
      --note "round-3 shipped build ($(cat $R/profiles/r03/HEAD.txt 2>/dev/null)): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) and --kernel-trace --stats over scripts/pmc_unet.py $cfg $rows --load-hints: UNet-only forwards at UNet batch $rows with the tiles the tuner pinned in the un-profiled run" > $OUT/pmc_${cfg}_summary.log 2>&1
  cp $OUT/pmc_${bc}_b${b}.json profiles/r03/
  find $OUT/trace_$cfg -name "*kernel_stats.csv" -exec cp {} $OUT/${cfg}_unet_only_kernel_stats.csv \;
  find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
}
echo "== PMC sd15 rows 16"; pmc_config sd15 16 sd15 8; grep -A12 '"igemm"' $OUT/pmc_sd15_b8.json | head -16
echo "== PMC sdxl rows 4"; pmc_config sdxl 4 sdxl 2; grep -A12 '"igemm"' $OUT/pmc_sdxl_b2.json | head -16
echo "== bench (default command: sd15 b8 + the SDXL b2 leg)"; timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
echo "== rocprofv3 --kernel-trace --stats of the bench command"
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/rocprof_bench -o sd15 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > $R/$OUT/bench_sd15_b8_under_rocprof.json 2> $R/$OUT/rocprof_bench.log; cd $R
find $OUT/rocprof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/sd15_b8_kernel_stats.csv \;
find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
head -14 $OUT/sd15_b8_kernel_stats.csv | cut -c1-170
echo "== per-launch profiles"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/unet_launch_profile_sd15_rows16.txt 2>&1; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/unet_launch_profile_sdxl_rows4.txt 2>&1
head -3 $OUT/unet_launch_profile_sd15_rows16.txt; head -3 $OUT/unet_launch_profile_sdxl_rows4.txt
du -sh $OUT
