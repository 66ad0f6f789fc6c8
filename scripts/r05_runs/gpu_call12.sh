#!/bin/bash
# Round-5 call 12 (FINAL build): whole GPU suite + smoke, then gpu_final.sh all (PMC + kernel-trace passes for the four bench
# populations, the bench lines, rocprofv3 --kernel-trace --stats of the bench command, per-launch tables, VAE profile)
set -u
OUT=gpurun_out/r05_call12; mkdir -p $OUT profiles/r05
export TMPDIR=/tmp CFGPP_BENCH_VERBOSE=0
git rev-parse HEAD > profiles/r05/HEAD.txt 2>/dev/null; cp profiles/r05/HEAD.txt $OUT/
rm -f gpurun_out/parity_r05.jsonl
timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $OUT/pytest_gpu.txt 2>&1; tail -14 $OUT/pytest_gpu.txt | cut -c1-300
cp gpurun_out/parity_r05.jsonl $OUT/parity_r05.jsonl 2>/dev/null
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/r05_runs/gpu_final.sh all
