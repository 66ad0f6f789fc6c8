#!/bin/bash
# Round-6 call 8: the persistent 256 x 256 kernels (configs 28 / 29): kernel parity, alone against the other tiles and the vendor
# (yardstick, token-major shapes), in situ (one engine per variant, each tuned for itself: candidates without / with 28 + 29)
set -u
OUT=gpurun_out/r06_call8; mkdir -p $OUT
export TMPDIR=/tmp
python -c "from cfgpp_amd import _lib; print(_lib.build_id())"
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > $OUT/pytest_kernels.log 2>&1; tail -5 $OUT/pytest_kernels.log | cut -c1-300
echo "== yardstick (token-major shapes)"
for f in GEGLU "QKV" "linear" "FF-out"; do timeout 300 python scripts/yardstick.py --filter "$f" --out $OUT/yardstick_$(echo $f | tr -d ' -').json 2>&1 | grep -v amdgpu.ids | grep -v wrote | cut -c1-230; done | tee $OUT/yardstick_token_major.txt
echo "== in situ sdxl b2"; timeout 900 python scripts/r06_runs/ab_forward.py sdxl 2 "base:mask=0xc1ffffff;persist:mask=0xf1ffffff;base2:mask=0xc1ffffff;persist2:mask=0xf1ffffff" > $OUT/ab_forward_persist_sdxl_b2.txt 2>&1; grep -E "^(base|persist)" $OUT/ab_forward_persist_sdxl_b2.txt | cut -c1-420
echo "== in situ sd15 b8"; timeout 900 python scripts/r06_runs/ab_forward.py sd15 8 "base:mask=0xc1ffffff;persist:mask=0xf1ffffff;base2:mask=0xc1ffffff;persist2:mask=0xf1ffffff" > $OUT/ab_forward_persist_sd15_b8.txt 2>&1; grep -E "^(base|persist)" $OUT/ab_forward_persist_sd15_b8.txt | cut -c1-420
