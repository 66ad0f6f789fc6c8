#!/bin/bash
# round-2 GPU run 4: quick validation (everything except the three minute-long real-size oracle runs) + A/B of the tile walk / deferred rescale
set -u
OUT=gpurun_out/r02_run4; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu (quick subset)"; timeout 1500 python -m pytest tests -m gpu -q -k "not real_" 2>&1 | tail -30 > $OUT/pytest_gpu_quick.txt; tail -8 $OUT/pytest_gpu_quick.txt
echo "== sd15 rows=16 forward vs oracle"; timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "real_unet_forward_at_bench_size and sd15" 2>&1 | tail -3
cp gpurun_out/parity_r02.jsonl $OUT/ 2>/dev/null
echo "== attention"
for a in "16 8 4096 40" "4 20 1024 64" "4 10 4096 64" "16 8 1024 80" "16 8 4096 40 77"; do ATTN_MODE=1 timeout 60 python scripts/one_attn.py $a 2>&1 | tail -1 | tee -a $OUT/attn.txt; done
echo "== geglu tile walk A/B (N_MAJOR env: -1 auto, 0 M-major)"
for nm in 0 -1; do for s in "geglu_m4096 10" "geglu_m4096 4" "conv_l2_1280 7" "lin_m4096_ffout 7"; do N_MAJOR=$nm timeout 60 python scripts/one_igemm.py $s 2>&1 | tail -1 | sed "s/^/n_major=$nm /" | tee -a $OUT/nmajor_ab.txt; done; done
echo "== profile sd15"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15.txt 2>&1; head -3 $OUT/prof_sd15.txt; grep "groupnorm HW=4096 C=320\|wall" $OUT/prof_sd15.txt
echo "== profile sdxl"; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl.txt 2>&1; head -3 $OUT/prof_sdxl.txt; tail -1 $OUT/prof_sdxl.txt
du -sh $OUT
