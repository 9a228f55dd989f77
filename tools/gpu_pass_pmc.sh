#!/bin/bash
# GRBM-level busy counters over the training kernels (one pass; TCC / TA counter sets crashed rocprofv3 on this pool)
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r02p
cd /tmp
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY -d $O/g1 -o t -- python $R/tools/clk_probe.py > $O/g1.log 2>&1 || tail -3 $O/g1.log
cd $R
python tools/pmc_table.py $O/g1 k_mlp_bwd k_mlp_fused k_mlp_pp | tee $O/table_grbm.txt
