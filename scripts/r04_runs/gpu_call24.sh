#!/bin/bash
# call 24: average shader clock over back-to-back UNet forwards (SD1.5 b8)
set -u
O=gpurun_out/r04_call24; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libclock_probe.so scripts/r04_runs/clock_probe.hip 2> $O/build.log || tail -5 $O/build.log
timeout 300 python scripts/r04_runs/forward_clock.py sd15 8 /tmp/libclock_probe.so 2>&1 | grep -v amdgpu.ids | tee $O/forward_clock_sd15_b8.txt
