#!/bin/bash
# call 27 (and 28, with the timing-only rows): loader-wave / MFMA-wave split of an LDS-DMA GEMM tile loop (micro_gemm_ws.hip), L2-resident operands
set -u
O=gpurun_out/r04_call32; mkdir -p $O
cd scripts/r04_runs && hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_gemm_ws.bin micro_gemm_ws.hip 2> ../../$O/build.log && timeout 120 /tmp/micro_gemm_ws.bin | tee ../../$O/micro_gemm_ws.txt
