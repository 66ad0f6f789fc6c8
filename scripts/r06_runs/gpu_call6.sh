#!/bin/bash
# Round-6 call 6: the yardstick again with forced tiles validated (a forced config bypasses the launcher's validity rules), the
# GEGLU shapes as plain GEMMs on our side, and the vendor kernels' names for the shapes where the vendor column is close or ahead.
set -u
OUT=gpurun_out/r06_call6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/yardstick.py --out $OUT/yardstick.json > $OUT/yardstick.txt 2>&1; grep -v amdgpu.ids $OUT/yardstick.txt | cut -c1-230
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace_vendor -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/yardstick.py --filter "GEGLU 32" --iters 5 --out /tmp/y.json > $GRAFT_REPO_ROOT/$OUT/trace_vendor.log 2>&1; cd $GRAFT_REPO_ROOT
find $OUT/trace_vendor -name "*kernel_stats.csv" -exec cp {} $OUT/vendor_kernel_stats_geglu32.csv \;
find $OUT -name "*.csv" -size +1M -delete 2>/dev/null; find $OUT -name "*.db" -delete 2>/dev/null
head -12 $OUT/vendor_kernel_stats_geglu32.csv | cut -c1-260
