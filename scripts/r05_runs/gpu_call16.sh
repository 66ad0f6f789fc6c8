#!/bin/bash
# Round-5 call 16 (last build: + the XCD-blocked walk): model-level GPU tests, per-launch-class PMC of the SDXL population again,
# the default bench line
set -u
OUT=gpurun_out/r05_call16; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
R=$GRAFT_REPO_ROOT
timeout 700 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py tests/test_gpu_step.py tests/test_gpu_weights.py tests/test_gpu_examples.py tests/test_gpu_text.py tests/test_gpu_torch_semantics.py -q -m gpu -p no:cacheprovider > $OUT/pytest_subset.txt 2>&1; tail -3 $OUT/pytest_subset.txt | cut -c1-400
export CFGPP_TUNE_CACHE=0
timeout 300 python scripts/pmc_unet.py sdxl 4 --save-hints > $OUT/sdxl_hints.log 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/sdxl_fetch -o p --output-format csv -- python $R/scripts/pmc_unet.py sdxl 4 --load-hints --iters 3 > $R/$OUT/sdxl_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/sdxl_write -o p --output-format csv -- python $R/scripts/pmc_unet.py sdxl 4 --load-hints --iters 3 > $R/$OUT/sdxl_write.log 2>&1
cd $R
python scripts/pmc_per_class.py --fetch $OUT/sdxl_fetch --write $OUT/sdxl_write --detail gpurun_out/detail_sdxl_rows4.txt --rows 4 > $OUT/pmc_per_launch_class_sdxl_rows4_blocked_walk.txt 2>&1
head -12 $OUT/pmc_per_launch_class_sdxl_rows4_blocked_walk.txt | cut -c1-150
find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
unset CFGPP_TUNE_CACHE
echo "== bench (default)"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json
