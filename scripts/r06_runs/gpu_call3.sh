#!/bin/bash
# Round-6 call 3: host-ASan pass over the sequence that preceded the round-5 abort; the two-wave attention workgroups: parity
# (both shapes forced, bit-identical) and the same-engine forward A/B.
set -u
OUT=gpurun_out/r06_call3; mkdir -p $OUT
export TMPDIR=/tmp
python -c "from cfgpp_amd import _lib; print(_lib.build_id())"
echo "== attention tests"; timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu -p no:cacheprovider -k "attention" > $OUT/pytest_attention.log 2>&1; tail -4 $OUT/pytest_attention.log | cut -c1-250
echo "== A/B sdxl b2"; timeout 600 python scripts/r06_runs/ab_attn_nw.py sdxl 2 4 > $OUT/ab_attn_nw_sdxl_b2.txt 2>&1; tail -4 $OUT/ab_attn_nw_sdxl_b2.txt | cut -c1-300
echo "== A/B sdxl b1"; timeout 600 python scripts/r06_runs/ab_attn_nw.py sdxl 1 4 > $OUT/ab_attn_nw_sdxl_b1.txt 2>&1; tail -4 $OUT/ab_attn_nw_sdxl_b1.txt | cut -c1-300
echo "== host ASan"; bash scripts/r06_runs/asan_repro.sh $OUT/asan
