#!/bin/bash
# Round-6 final GPU run ON THE SHIPPED BUILD (profiles/r06/HEAD.txt): PMC passes (FETCH / WRITE / SQ, each alone) + kernel trace over
# UNet-only forwards of the four bench populations (sd15 rows 16, sdxl rows 4, sdxl rows 16 = Lightning b8, sdxl rows 2 = edit b1),
# the default bench command (SD1.5 b8 + the SDXL b2 leg + cpu_baseline), the Lightning and edit bench lines, rocprofv3 --kernel-trace
# --stats of the bench command, per-launch tables, VAE profile.   usage: gpu_final.sh [pmc|bench|all]
set -u
WHAT=${1:-all}
OUT=gpurun_out/r06_final; mkdir -p $OUT profiles/r06
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0 CFGPP_TUNE_CACHE=0
R=$GRAFT_REPO_ROOT
HEAD=$(python -c "from cfgpp_amd import _lib; print(_lib.build_id())" 2>/dev/null | tail -1)
pmc_config() {   # unet_config rows bench_config batch
  local cfg=$1 rows=$2 bc=$3 b=$4
  timeout 400 python scripts/pmc_unet.py $cfg $rows --save-hints > $OUT/pmc_${bc}_hints.log 2>&1
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_$bc -o t --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 10 > $R/$OUT/trace_$bc.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_${bc}_fetch -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_${bc}_write -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_write.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/$OUT/pmc_${bc}_sq -o p --output-format csv -- python $R/scripts/pmc_unet.py $cfg $rows --load-hints --iters 3 > $R/$OUT/pmc_${bc}_sq.log 2>&1
  cd $R
  python scripts/pmc_summary.py --fetch $OUT/pmc_${bc}_fetch --write $OUT/pmc_${bc}_write --sq $OUT/pmc_${bc}_sq --trace $OUT/trace_$bc \
      --detail gpurun_out/detail_${cfg}_rows${rows}.txt --rows $rows --out $OUT/pmc_${bc}_b${b}.json \
      --note "round-6 shipped build ($HEAD): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes) and --kernel-trace --stats over scripts/pmc_unet.py $cfg $rows --load-hints: UNet-only forwards at UNet batch $rows with the tiles the tuner pinned in the un-profiled run" > $OUT/pmc_${bc}_summary.log 2>&1
  cp $OUT/pmc_${bc}_b${b}.json profiles/r06/
  find $OUT/trace_$bc -name "*kernel_stats.csv" -exec cp {} $OUT/${bc}_unet_only_kernel_stats.csv \;
  find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
}
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  echo "== PMC sd15 rows 16"; pmc_config sd15 16 sd15 8; grep -A14 '"igemm": {' $OUT/pmc_sd15_b8.json | head -18
  echo "== PMC sdxl rows 4"; pmc_config sdxl 4 sdxl 2; grep -A14 '"igemm": {' $OUT/pmc_sdxl_b2.json | head -18
  echo "== PMC sdxl rows 16 (Lightning b8)"; pmc_config sdxl 16 sdxl_lightning 8; grep -A6 '"igemm": {' $OUT/pmc_sdxl_lightning_b8.json | head -8
  echo "== PMC sdxl rows 2 (edit b1)"; pmc_config sdxl 2 sdxl_edit 1; grep -A6 '"igemm": {' $OUT/pmc_sdxl_edit_b1.json | head -8
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  echo "== bench (default command: sd15 b8 + the SDXL b2 leg)"; timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json
  echo "== bench sdxl_lightning b8"; timeout 900 python bench.py --config sdxl_lightning --no-cpu-baseline > $OUT/bench_sdxl_lightning_b8.json 2> $OUT/bench_lightning.err; cat $OUT/bench_sdxl_lightning_b8.json | cut -c1-600
  echo "== bench sdxl_edit b1"; timeout 900 python bench.py --config sdxl_edit --no-cpu-baseline > $OUT/bench_sdxl_edit_b1.json 2> $OUT/bench_edit.err; cat $OUT/bench_sdxl_edit_b1.json | cut -c1-600
  echo "== rocprofv3 --kernel-trace --stats of the bench command"
  cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/rocprof_bench -o sd15 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also > $R/$OUT/bench_sd15_b8_under_rocprof.json 2> $R/$OUT/rocprof_bench.log; cd $R
  find $OUT/rocprof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/sd15_b8_kernel_stats.csv \;
  find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
  head -14 $OUT/sd15_b8_kernel_stats.csv | cut -c1-170
  echo "== per-launch profiles"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/unet_launch_profile_sd15_rows16.txt 2>&1; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/unet_launch_profile_sdxl_rows4.txt 2>&1
  head -3 $OUT/unet_launch_profile_sd15_rows16.txt; head -3 $OUT/unet_launch_profile_sdxl_rows4.txt
  timeout 300 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64.txt 2>&1; head -3 $OUT/vae_b8_64.txt
fi
du -sh $OUT
