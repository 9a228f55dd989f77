#!/bin/bash
# PMC passes over the training step: where do the stores of the training kernels stall?
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r02p
cd /tmp
i=0
for set in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d $O/p$i -o t -- python $R/tools/train_trace.py 3 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
cd $R
python tools/pmc_table.py $O | tee $O/table.txt
