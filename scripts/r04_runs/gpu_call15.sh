#!/bin/bash
# call 15: issue-logic micro-benchmarks for the attention redesign (no library code involved)
#   micro_overlap: matrix pipe || softmax VALU on one CU at 1-4 waves per SIMD; micro_valu: + v_max3_f32, v_cvt_pk_f16_f32, packed fp16
set -u
O=gpurun_out/r04_call15; mkdir -p $O
cd scripts/r04_runs
for b in micro_overlap micro_valu; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/$b.bin $b.hip 2> ../../$O/$b.build.log || { echo "build of $b failed"; tail -5 ../../$O/$b.build.log; continue; }
  for rep in 1 2; do timeout 120 /tmp/$b.bin | tee ../../$O/$b.run$rep.txt; done
done
