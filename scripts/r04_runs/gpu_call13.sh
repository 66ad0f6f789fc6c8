#!/bin/bash
# Round-4 call 13: sanity of the last commit's binary (rebuilt after comment-only changes): smoke(), kernel + step tests
set -u
OUT=gpurun_out/r04_call13; mkdir -p $OUT
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_torch_semantics.py -q -m gpu > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
