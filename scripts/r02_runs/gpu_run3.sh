#!/bin/bash
# round-2 GPU run 3: XCD-aware attention order, GEGLU 256x320 tile, GroupNorm apply prologue, VAE fix; PMC wave-time splits
set -u
OUT=gpurun_out/r02_run3; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu" ; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -12 $OUT/pytest_gpu.txt
cp gpurun_out/parity_r02.jsonl $OUT/ 2>/dev/null
echo "== attention A/B (mode 0 = register staged, 1 = DMA 3-stage)"
for m in 0 1; do for a in "16 8 4096 40" "4 20 1024 64" "4 10 4096 64" "16 8 1024 80" "16 8 4096 40 77" "4 20 1024 64 77"; do ATTN_MODE=$m timeout 60 python scripts/one_attn.py $a 2>&1 | tail -1 | sed "s/^/mode$m /" | tee -a $OUT/attn_ab.txt; done; done
echo "== igemm geglu / M=4096"
for s in "geglu_m4096 0" "geglu_m4096 4" "geglu_m4096 10" "geglu_m4096 1" "geglu_l0 0" "geglu_l0 4" "geglu_l0 10" "lin_m4096_1280 7" "lin_m4096_1280 1" "lin_m4096_ffout 7"; do timeout 60 python scripts/one_igemm.py $s 2>&1 | tail -1 | tee -a $OUT/igemm_one.txt; done
echo "== PMC wave-time split"
cd /tmp
PMCA="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
PMCB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS"
PMCC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
i=0
for t in "attn:16 8 4096 40 4096 5" "igemm:lin_m4096_1280 7 5" "igemm:conv_l2_1280 7 5" "igemm:geglu_m4096 10 5" "igemm:conv_l0_640in 5 5"; do
  kind=${t%%:*}; args=${t#*:}; i=$((i+1)); tag=${kind}_$i
  if [ $kind = attn ]; then cmd="python $R/scripts/one_attn.py $args"; else cmd="python $R/scripts/one_igemm.py $args"; fi
  timeout 150 rocprofv3 --pmc $PMCA -d $R/$OUT/pmc_${tag}_a -o pmc --output-format csv -- $cmd > $R/$OUT/pmc_${tag}_a.log 2>&1
  timeout 150 rocprofv3 --pmc $PMCB -d $R/$OUT/pmc_${tag}_b -o pmc --output-format csv -- $cmd > $R/$OUT/pmc_${tag}_b.log 2>&1
  timeout 150 rocprofv3 --pmc $PMCC -d $R/$OUT/pmc_${tag}_c -o pmc --output-format csv -- $cmd > $R/$OUT/pmc_${tag}_c.log 2>&1
  python $R/scripts/pmc_sq.py $R/$OUT/pmc_${tag}_a $R/$OUT/pmc_${tag}_b $R/$OUT/pmc_${tag}_c 2>/dev/null | grep -A24 "igemm_kernel\|attn" | head -30 > $R/$OUT/pmc_${tag}.txt
  echo "--- $t"; cat $R/$OUT/pmc_${tag}.txt
done
cd $R
find $OUT -name "*.csv" -size +2M -delete 2>/dev/null
echo "== profile sd15"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15.txt 2>&1; head -3 $OUT/prof_sd15.txt; tail -2 $OUT/prof_sd15.txt
echo "== profile sdxl"; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl.txt 2>&1; head -3 $OUT/prof_sdxl.txt; tail -2 $OUT/prof_sdxl.txt
echo "== bench sd15"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_sd15.json 2> $OUT/bench_sd15.err; cat $OUT/bench_sd15.json
du -sh $OUT
