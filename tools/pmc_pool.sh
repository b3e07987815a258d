#!/bin/bash
# Run ON THE GPU BOX: PMC counters of the conv-pool kernels of one MViTv2-S block (tools/probe/mvit_pool_times.py), one
# rocprofv3 pass per counter group; prints the mean per kernel.  usage: tools/pmc_pool.sh [block]
export TMPDIR=/tmp
R=$PWD
B=${1:-4}
cd /tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  PVRL_POOL_BLOCKS=$B rocprofv3 --kernel-trace --pmc $c -d /tmp/pp/g$i -o pmc --output-format csv -- python $R/tools/probe/mvit_pool_times.py > /tmp/pp_$i.log 2>&1
done
find /tmp/pp -name "*counter_collection.csv" | head
python $R/tools/pmc_agg.py /tmp/pp pool 2>&1 | grep pool
