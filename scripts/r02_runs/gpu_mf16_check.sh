#!/bin/bash
set -u
OUT=gpurun_out/r02_mf16; mkdir -p $OUT
timeout 100 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "8wave" 2>&1 | tail -15 | tee $OUT/pytest_mf16.txt
timeout 20 python - <<'PY' 2>&1 | tail -12 | tee -a $OUT/pytest_mf16.txt
import sys, json; sys.path.insert(0, "tests")
import gpu_diag as d
d.t_big()
r = d.RESULTS["igemm_big_tiles"]
for k, v in r.items():
    if "cfg18" in k or "cfg19" in k or "cfg7" in k: print(k, {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items() if a in ("rel_l2", "max_abs", "finite", "halo_zero")})
PY
