#!/bin/bash
set -u
OUT=gpurun_out/r06_call5; mkdir -p $OUT
export TMPDIR=/tmp
echo "== ASan, SEGV handler on (where does HIP init die under the ASan runtime?)"
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
( export CFGPP_LIB=$PWD/cfgpp_amd/libcfgpp_hip_asan.so ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:allocator_may_return_null=1
  LD_PRELOAD=$ASAN timeout 300 python -c "import torch; print(torch.cuda.is_available()); print(torch.zeros(4).cuda().sum().item())" > $OUT/asan_hip_init.log 2>&1; echo "rc=$?"; tail -25 $OUT/asan_hip_init.log | cut -c1-200 )
echo "== glibc malloc checker over the repro sequence (shipped library)"
bash scripts/r06_runs/malloc_check_repro.sh $OUT
