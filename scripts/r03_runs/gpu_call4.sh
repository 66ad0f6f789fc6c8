#!/bin/bash
# Round 3, GPU call 4: LayerNorm folded into the consuming projections, the resident-K/V cross-attention kernel, the
# software-pipelined attention kernel and the phase-stagger knob: correctness, single-kernel A/Bs, in-situ A/Bs.
set -u
OUT=gpurun_out/r03_call4; mkdir -p $OUT
export CFGPP_BENCH_VERBOSE=0
echo "== 1 targeted tests"
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_torch_semantics.py -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_kernels.txt
timeout 500 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "attention" 2>&1 | tail -12 | tee $OUT/pytest_attention.txt
timeout 500 python -m pytest tests/test_gpu_unet.py -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_unet.txt
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "real_unet_forward_at_bench_size" 2>&1 | tail -6 | tee $OUT/pytest_sd15_forward.txt
echo "== 2 attention kernels alone"
att() { ATTN_MODE=$1 ATTN_STAGGER=$2 ATTN_CROSS=$3 timeout 60 python scripts/one_attn.py ${@:4} 2>&1 | grep "^attn"; }
for sh in "16 8 4096 40" "4 10 4096 64" "4 20 1024 64"; do
  att 1 0 1 $sh; att 1 2 1 $sh; att 1 4 1 $sh; att 1 8 1 $sh; att 2 0 1 $sh; att 2 4 1 $sh
done 2>&1 | tee $OUT/attn_self.txt
for sh in "16 8 4096 40 77" "4 20 1024 64 77" "4 10 4096 64 77"; do
  att 1 0 0 $sh; att 1 0 1 $sh
done 2>&1 | tee $OUT/attn_cross.txt
echo "== 3 in-situ"
prof() { name=$1; shift; env "$@" timeout 200 python scripts/profile_unet.py ${CFG} > $OUT/prof_${CFGN}_$name.txt 2>&1; echo "$name: $(grep '^# ' $OUT/prof_${CFGN}_$name.txt | head -2 | tr '\n' ' ')"; }
for c in "sd15 16" "sdxl 4"; do
  CFG="$c"; CFGN=$(echo $c | tr ' ' '_')
  echo "-- $c"
  prof base FUSE_LN=0 ATTN_CROSS=0
  prof ln FUSE_LN=1 ATTN_CROSS=0
  prof ln_x FUSE_LN=1 ATTN_CROSS=1
  prof ln_x_stag4 FUSE_LN=1 ATTN_CROSS=1 ATTN_STAGGER=4
  prof ln_x_pipe FUSE_LN=1 ATTN_CROSS=1 ATTN_MODE=2
done
cp gpurun_out/parity_r03.jsonl $OUT/ 2>/dev/null
du -sh $OUT
