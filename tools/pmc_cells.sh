#!/bin/bash
# PMC view of the data-gradient pass in the four cells of the weight stream x stores ablation (build/ab/libpnr_{head,nostore,nodma,neither}.so,
# tools/build_ab.sh), one rocprofv3 --pmc pass per counter group and cell; unknown counter names fail their pass only.
#   usage (on the GPU box): tools/pmc_cells.sh <out dir>
R=$(pwd); O=$R/${1:-gpurun_out/pmc_cells}; mkdir -p $O; export TMPDIR=/tmp
CG=("TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN2_sum"
        "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"
        "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES"
        "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCC_BUSY_avr")
for cell in head nostore nodma neither; do
  g=0
  for grp in "${CG[@]}"; do
    cd /tmp
    PNR_LIB_PATH=$R/build/ab/libpnr_$cell.so timeout 120 rocprofv3 --pmc $grp -d $O/db_${cell}_$g -o t -- python $R/tools/train_kernels_time.py 4096 > $O/run_${cell}_$g.log 2>&1
    cd $R
    db=$(find $O/db_${cell}_$g -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/pmc_dump.py $db --like k_mlp_bwd --like k_mlp_fused >> $O/pmc_$cell.txt 2>/dev/null; else echo "group $g: no database (a counter name this part does not have?)" >> $O/pmc_$cell.txt; tail -3 $O/run_${cell}_$g.log >> $O/pmc_$cell.txt; fi
    rm -rf $O/db_${cell}_$g
    g=$((g+1))
  done
done
wc -l $O/pmc_*.txt
