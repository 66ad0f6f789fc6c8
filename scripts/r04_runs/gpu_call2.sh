#!/bin/bash
# Round-4 call 2: first contact of lin32_kernel (token-major linears on 32-deep K-tiles, 2-3 workgroups per CU): kernel tests,
# then same-box forward A/Bs with the new tiles masked out / offered to the tuner / offered with the 16x16x32 rule restricted to convs.
set -u
OUT=gpurun_out/r04_call2; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
echo "== kernel tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -5 $OUT/pytest_kernels.txt
python - <<'PY' > $OUT/lin32_results.txt 2>&1
import json, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import gpu_diag
gpu_diag.t_lin32()
for k, v in gpu_diag.RESULTS["lin32_unet_sizes"].items(): print(k, json.dumps(v))
PY
cat $OUT/lin32_results.txt | cut -c1-220
NOLIN=0xfffc7fff   # tuner mask without configs 15 / 16 / 17
echo "== A/B sd15 b8"; timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "base:mask=$NOLIN;lin32:mask=0xffffffff;lin32_mf16convonly:mask=0xffffffff,mf16lin=0" --table > $OUT/ab_sd15_b8.txt 2>&1; head -5 $OUT/ab_sd15_b8.txt | cut -c1-400
echo "== A/B sdxl b2"; timeout 1500 python scripts/r04_runs/ab_forward.py sdxl 2 "base:mask=$NOLIN;lin32:mask=0xffffffff;lin32_mf16convonly:mask=0xffffffff,mf16lin=0" --table > $OUT/ab_sdxl_b2.txt 2>&1; head -5 $OUT/ab_sdxl_b2.txt | cut -c1-400
echo "== tables"; sed -n 5,60p $OUT/ab_sd15_b8.txt
echo "== model-level parity (tuner now offers lin32)"
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "forward_vs_oracle or check_finite" > $OUT/pytest_unet.txt 2>&1; tail -3 $OUT/pytest_unet.txt
