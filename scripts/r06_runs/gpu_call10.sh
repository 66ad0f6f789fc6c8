#!/bin/bash
# Round-6 call 10 (shipped build): the persistent tile in situ at the other two SDXL populations (Lightning batch 8 = 16 rows,
# edit batch 1 = 2 rows), and the full yardstick with config 28 among the candidate tiles
set -u
OUT=gpurun_out/r06_call10; mkdir -p $OUT
export TMPDIR=/tmp
python -c "from cfgpp_amd import _lib; print(_lib.build_id())"
echo "== in situ sdxl b8 (Lightning rows 16)"; timeout 1200 python scripts/r06_runs/ab_forward.py sdxl 8 "base:mask=0xe1ffffff;persist:mask=0xf1ffffff;base2:mask=0xe1ffffff;persist2:mask=0xf1ffffff" > $OUT/ab_forward_persist_sdxl_b8.txt 2>&1; grep -E "^(base|persist)" $OUT/ab_forward_persist_sdxl_b8.txt | cut -c1-420
echo "== in situ sdxl b1 (edit rows 2)"; timeout 900 python scripts/r06_runs/ab_forward.py sdxl 1 "base:mask=0xe1ffffff;persist:mask=0xf1ffffff;base2:mask=0xe1ffffff;persist2:mask=0xf1ffffff" > $OUT/ab_forward_persist_sdxl_b1.txt 2>&1; grep -E "^(base|persist)" $OUT/ab_forward_persist_sdxl_b1.txt | cut -c1-420
echo "== yardstick"; timeout 900 python scripts/yardstick.py --out $OUT/yardstick.json > $OUT/yardstick.txt 2>&1; grep -v amdgpu.ids $OUT/yardstick.txt | cut -c1-230
