#!/bin/bash
# round-2 GPU run 2: full parity suite on the new kernels + attention / GroupNorm / igemm A/B + tuned profiles + bench
set -u
OUT=gpurun_out/r02_run2; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" ; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest_gpu.txt; tail -15 $OUT/pytest_gpu.txt
cp gpurun_out/parity_r02.jsonl $OUT/ 2>/dev/null
echo "== attention A/B (mode 0 = register staged, 1 = DMA 3-stage, 2 = DMA 2-stage)"
for m in 0 1 2; do for a in "16 8 4096 40" "4 20 1024 64" "4 10 4096 64" "16 8 4096 40 77" "4 20 1024 64 77"; do ATTN_MODE=$m timeout 60 python scripts/one_attn.py $a 2>&1 | tail -1 | sed "s/^/mode$m /" | tee -a $OUT/attn_ab.txt; done; done
echo "== groupnorm A/B"; timeout 200 python scripts/bench_norm.py 2>&1 | grep "^gn" | tee $OUT/bench_norm.txt
echo "== igemm M=4096 shapes"; CFGS=0,1,7,9 timeout 200 python scripts/bench_igemm.py _32 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_igemm_m4096.txt; CFGS=0,1,7,9 timeout 100 python scripts/bench_igemm.py l2_ 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench_igemm_m4096.txt
echo "== profile sd15"; timeout 300 python scripts/profile_unet.py sd15 16 > $OUT/prof_sd15.txt 2>&1; head -3 $OUT/prof_sd15.txt; tail -2 $OUT/prof_sd15.txt
echo "== profile sdxl"; timeout 400 python scripts/profile_unet.py sdxl 4 > $OUT/prof_sdxl.txt 2>&1; head -3 $OUT/prof_sdxl.txt; tail -2 $OUT/prof_sdxl.txt
echo "== bench sd15"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_sd15.json 2> $OUT/bench_sd15.err; cat $OUT/bench_sd15.json
du -sh $OUT
