#!/bin/bash
# Round-6 call 9: the shipped build with the persistent tile: the driver's exact GPU-suite command, then the final measurement set again
set -u
OUT=gpurun_out/r06_call9; mkdir -p $OUT
export TMPDIR=/tmp
python -c "from cfgpp_amd import _lib; print(_lib.build_id())" | tee $OUT/build_id.txt
t0=$(date +%s)
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=8 > $OUT/pytest_driver_cmd.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"; tail -14 $OUT/pytest_driver_cmd.log | cut -c1-200
cp gpurun_out/parity_realsize_*.jsonl $OUT/ 2>/dev/null
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r06_runs/gpu_final.sh all
