#!/bin/bash
# call 21: sustained clock under matrix-pipe load (micro_clock.hip)
set -u
O=gpurun_out/r04_call21; mkdir -p $O
cd scripts/r04_runs && hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_clock.bin micro_clock.hip 2> ../../$O/build.log && timeout 120 /tmp/micro_clock.bin | tee ../../$O/micro_clock.txt
