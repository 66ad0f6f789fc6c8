#!/bin/bash
# Round-6 call 1: reproduce the round-5 driver abort (rc 134 between tests #22 and #23 of the alphabetical order) with the
# driver's exact command on the HEAD build, diagnostics on, the log kept whole.
set -u
OUT=gpurun_out/r06_call1; mkdir -p $OUT
export TMPDIR=/tmp
free -g > $OUT/box.txt; nproc >> $OUT/box.txt; ulimit -a >> $OUT/box.txt; cat /sys/fs/cgroup/memory.max >> $OUT/box.txt 2>&1
( export CFGPP_TEST_ORDER=alpha PYTHONFAULTHANDLER=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 AMD_LOG_LEVEL=1
  timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -v --ignore=tests/test_gpu_realsize.py > $OUT/pytest_driver_cmd.log 2>&1; echo "rc=$?" > $OUT/pytest_driver_cmd.rc )
cat $OUT/pytest_driver_cmd.rc
head -c 6000 $OUT/pytest_driver_cmd.log > $OUT/pytest_head.txt
grep -n -i -E "abort|corrupt|malloc|free\(\)|HSA|fault|terminate|what\(\)|double|invalid|smash|Fatal|core" $OUT/pytest_driver_cmd.log | head -40
grep -n -E "PASSED|FAILED|ERROR" $OUT/pytest_driver_cmd.log | tail -5
dmesg 2>/dev/null | tail -20 > $OUT/dmesg.txt
# the two tests alone, in the order that died, each with its own log
( export CFGPP_TEST_ORDER=alpha PYTHONFAULTHANDLER=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 AMD_LOG_LEVEL=1
  timeout 900 python3 -m pytest "tests/test_gpu_configs.py::test_real_unet_forward_at_bench_size" "tests/test_gpu_configs.py::test_real_sdxl_forward_at_every_bench_plan_size" -x -q -p no:cacheprovider -v > $OUT/pytest_pair.log 2>&1; echo "pair rc=$?" )
tail -5 $OUT/pytest_pair.log | cut -c1-300
free -g >> $OUT/box.txt
