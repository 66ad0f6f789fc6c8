#!/bin/bash
# glibc heap checker (libc_malloc_debug.so: on glibc >= 2.34 MALLOC_CHECK_ only works with it preloaded - call 1's MALLOC_CHECK_=3
# alone was a no-op) over the sequence that preceded the round-5 abort, with the SHIPPED library.  Detects writes past a heap
# block (canary checked at free / realloc), double frees, frees of foreign pointers; aborts with a message when it finds one.
set -u
OUT=${1:-gpurun_out/r06_malloc_check}; mkdir -p $OUT
export CFGPP_TUNE_CACHE=0 PYTHONFAULTHANDLER=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165
LD_PRELOAD=/lib/x86_64-linux-gnu/libc_malloc_debug.so.0 timeout 1500 python scripts/r06_runs/asan_repro.py > $OUT/malloc_check_repro.log 2>&1
echo "malloc-check repro rc=$?"; tail -12 $OUT/malloc_check_repro.log | cut -c1-250
