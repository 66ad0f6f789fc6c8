#!/bin/bash
set -u
OUT=gpurun_out/r02_mf16; mkdir -p $OUT
timeout 95 python - <<'PY' 2>&1 | tail -8 | tee $OUT/pytest_mf16_unet.txt
import sys, pytest
from cfgpp_amd import _lib
_lib.load().cfgpp_igemm_set_mf16(4)          # the 16x16x32-MFMA tile by rule, then the real SD1.5 16-row forward vs the fp32 oracle
sys.exit(pytest.main(["tests/test_gpu_configs.py", "-m", "gpu", "-x", "-q", "-k", "real_unet_forward and sd15"]))
PY
grep real_unet gpurun_out/parity_r02.jsonl | tail -1 | tee -a $OUT/pytest_mf16_unet.txt
