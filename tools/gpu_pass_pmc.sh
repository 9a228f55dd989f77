#!/bin/bash
# SQ-level LDS counters of the inference MLP (ping-pong and lock-step forms): is the LDS the busy unit?
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r02p
cd /tmp
for var in 1 0; do
  PNR_MLP_VARIANT=$var timeout 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/lds$var -o t -- python $R/tools/mlp_once.py 2 > $O/lds$var.log 2>&1 || tail -3 $O/lds$var.log
  PNR_MLP_VARIANT=$var timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/ldsb$var -o t -- python $R/tools/mlp_once.py 2 > $O/ldsb$var.log 2>&1 || tail -3 $O/ldsb$var.log
done
cd $R
for var in 1 0; do echo "#### PNR_MLP_VARIANT=$var"; python tools/pmc_table.py $O/lds$var k_mlp_pp k_mlp_fused; python tools/pmc_table.py $O/ldsb$var k_mlp_pp k_mlp_fused; done | tee $O/table_lds.txt
rm -rf $O/lds0 $O/lds1 $O/ldsb0 $O/ldsb1
