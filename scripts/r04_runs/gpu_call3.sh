#!/bin/bash
# Round-4 call 3: first contact of rb_kernel (row-block linears, LayerNorm inside) and of the LDS-tiled conv_out: kernel tests,
# model-level parity, same-box forward A/Bs, VAE per-launch profile.
set -u
OUT=gpurun_out/r04_call3; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
echo "== kernel tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.txt 2>&1; tail -8 $OUT/pytest_kernels.txt
python - <<'PY' > $OUT/rb_results.txt 2>&1
import json, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import gpu_diag
gpu_diag.t_rb()
for k, v in gpu_diag.RESULTS["rowblock"].items(): print(k, json.dumps(v) if isinstance(v, dict) else v)
gpu_diag.t_cio()
for k, v in gpu_diag.RESULTS["conv_in_out"].items(): print(k, json.dumps(v) if isinstance(v, dict) else v)
PY
grep -v "^\[diag\]" $OUT/rb_results.txt | cut -c1-200
NO20=0xffefffff   # tuner mask without config 20
echo "== A/B sd15 b8"; timeout 900 python scripts/r04_runs/ab_forward.py sd15 8 "base:mask=$NO20,rbln=0;rb_tile_only:rbln=0;rb_ln:rbln=1" --table > $OUT/ab_sd15_b8.txt 2>&1; head -5 $OUT/ab_sd15_b8.txt | cut -c1-420
echo "== A/B sdxl b2"; timeout 1500 python scripts/r04_runs/ab_forward.py sdxl 2 "base:mask=$NO20,rbln=0;rb_ln:rbln=1" > $OUT/ab_sdxl_b2.txt 2>&1; head -5 $OUT/ab_sdxl_b2.txt | cut -c1-420
echo "== table"; sed -n 5,50p $OUT/ab_sd15_b8.txt
echo "== model-level parity"
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_vae.py -x -q -m gpu > $OUT/pytest_unet_vae.txt 2>&1; tail -3 $OUT/pytest_unet_vae.txt
echo "== VAE per-launch profile"; timeout 300 python scripts/profile_vae.py 8 64 > $OUT/vae_b8_64.txt 2>&1; head -12 $OUT/vae_b8_64.txt
