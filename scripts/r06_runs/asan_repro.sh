#!/bin/bash
# host-ASan pass (see asan_repro.py); usage: bash scripts/r06_runs/asan_repro.sh [outdir]
set -u
OUT=${1:-gpurun_out/r06_asan}; mkdir -p $OUT
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export CFGPP_LIB=$PWD/cfgpp_amd/libcfgpp_hip_asan.so CFGPP_TUNE_CACHE=0 PYTHONFAULTHANDLER=1
# leaks: the interpreter and torch never free at exit; alloc/dealloc mismatch + new-delete size checks stay on
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:protect_shadow_gap=0:handle_segv=0:allocator_may_return_null=1:log_path=$PWD/$OUT/asan
LD_PRELOAD=$ASAN timeout 1500 python scripts/r06_runs/asan_repro.py > $OUT/asan_repro.log 2>&1
echo "asan repro rc=$?"; tail -12 $OUT/asan_repro.log | cut -c1-250
ls $OUT; for f in $OUT/asan.*; do [ -f "$f" ] && head -60 "$f"; done
