#!/bin/bash
# usage: pmc_gemm2.sh M N K tag mode   (two counter passes: LDS / VMEM pressure)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$4
mkdir -p $OUT/a $OUT/b
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/scratch/gemm_one.py $1 $2 $3 $5 > $OUT/a/log.txt 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/scratch/gemm_one.py $1 $2 $3 $5 > $OUT/b/log.txt 2>&1
