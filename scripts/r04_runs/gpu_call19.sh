#!/bin/bash
# call 19: PMC pass over the attention kernels alone: flash loop, woven kernel, woven with parts switched off
set -u
R=$(pwd); O=gpurun_out/r04_call19; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
   -d /tmp/pmc_attn -o p --output-format csv -- python $R/scripts/r04_runs/pmc_attention.py run > $R/$O/pmc_run.log 2>&1
cd $R
python scripts/r04_runs/pmc_attention.py sum /tmp/pmc_attn 2>&1 | tee $O/attention_pmc.txt
tail -3 $O/pmc_run.log
