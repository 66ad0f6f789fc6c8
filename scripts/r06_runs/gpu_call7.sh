#!/bin/bash
# Round-6 call 7: final measurements on the shipped build (cfgpp_build_id printed first): PMC passes of the two default-bench
# populations + Lightning + edit, the default bench command, rocprofv3 --kernel-trace --stats of it, per-launch tables.
set -u
python -c "from cfgpp_amd import _lib; print(_lib.build_id())" | tee gpurun_out/r06_final_build_id.txt
bash scripts/r06_runs/gpu_final.sh all
cp gpurun_out/r06_final/*.json gpurun_out/r06_final/*.csv gpurun_out/r06_final/*.txt profiles/r06/ 2>/dev/null
