#!/bin/bash
# round-2 GPU run 1: new parity tests + igemm tile-family A/B + attention PMC baseline + tuned per-launch profiles
set -u
OUT=gpurun_out/r02_run1; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt | tail -8
echo "== bench_igemm"; timeout 400 python scripts/bench_igemm.py > $OUT/bench_igemm.txt 2>&1; tail -40 $OUT/bench_igemm.txt
echo "== profile sd15"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15.txt 2>&1; head -3 $OUT/prof_sd15.txt; tail -2 $OUT/prof_sd15.txt
echo "== profile sdxl"; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl.txt 2>&1; head -3 $OUT/prof_sdxl.txt; tail -2 $OUT/prof_sdxl.txt
echo "== attention timing"
for a in "16 8 4096 40" "4 20 1024 64" "4 10 4096 64" "16 8 1024 80" "16 8 4096 40 77" "4 20 1024 64 77"; do timeout 60 python scripts/one_attn.py $a 2>&1 | tail -1 | tee -a $OUT/attn_timing.txt; done
echo "== attention PMC (SQ)"
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/rocprof_counters.txt 2>&1 || true
for a in "16 8 4096 40" "4 20 1024 64"; do
  tag=$(echo $a | tr ' ' '_')
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $GRAFT_REPO_ROOT/$OUT/pmc_attn_a_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/one_attn.py $a 5 > $GRAFT_REPO_ROOT/$OUT/pmc_attn_a_$tag.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$OUT/pmc_attn_b_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/one_attn.py $a 5 > $GRAFT_REPO_ROOT/$OUT/pmc_attn_b_$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_sq.py $OUT/pmc_attn_a_* $OUT/pmc_attn_b_* > $OUT/pmc_attn_summary.txt 2>&1; grep -A12 "attn_kernel" $OUT/pmc_attn_summary.txt | head -60
# keep only the csv summaries small
find $OUT -name "*.db" -delete 2>/dev/null; du -sh $OUT
