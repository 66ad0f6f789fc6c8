#!/bin/bash
# Round-6 call 2: the driver's exact command on the re-ordered, fixture-based suite (new build with cfgpp_build_id and the
# whole-loop graph replay), then the vendor yardstick and the graph A/B.
set -u
OUT=gpurun_out/r06_call2; mkdir -p $OUT profiles/r06
export TMPDIR=/tmp
python -c "from cfgpp_amd import _lib; print(_lib.build_id())" > $OUT/build_id.txt 2>&1; cat $OUT/build_id.txt
t0=$(date +%s)
( export PYTHONFAULTHANDLER=1
  timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=15 > $OUT/pytest_driver_cmd.log 2>&1; echo "rc=$?" > $OUT/pytest_driver_cmd.rc )
echo "suite: $(cat $OUT/pytest_driver_cmd.rc) in $(( $(date +%s) - t0 )) s"; tail -25 $OUT/pytest_driver_cmd.log | cut -c1-250
cat gpurun_out/parity_realsize_*.jsonl 2>/dev/null | cut -c1-400
echo "== yardstick"; timeout 900 python scripts/yardstick.py --out $OUT/yardstick.json > $OUT/yardstick.txt 2>&1; cat $OUT/yardstick.txt | cut -c1-220
echo "== graph A/B sd15 b8 50 NFE"; timeout 600 python scripts/r06_runs/ab_graph.py sd15 8 50 4 > $OUT/ab_graph_sd15_b8.txt 2>&1; tail -5 $OUT/ab_graph_sd15_b8.txt | cut -c1-300
echo "== graph A/B sdxl b2 50 NFE"; timeout 600 python scripts/r06_runs/ab_graph.py sdxl 2 50 3 > $OUT/ab_graph_sdxl_b2.txt 2>&1; tail -5 $OUT/ab_graph_sdxl_b2.txt | cut -c1-300
