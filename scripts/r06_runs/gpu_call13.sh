#!/bin/bash
# Round-6 call 13: the final tree, exactly what the driver runs at round end: GPU suite, smoke(), default bench
set -u
OUT=gpurun_out/r06_call13; mkdir -p $OUT
python -c "from cfgpp_amd import _lib; print(_lib.build_id())" | tee $OUT/build_id.txt
t0=$(date +%s)
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_driver_cmd.log 2>&1; echo "suite rc=$? in $(( $(date +%s) - t0 )) s"; tail -4 $OUT/pytest_driver_cmd.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
