#!/bin/bash
# usage: pmc_gemm.sh M N K tag mode
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$4
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scratch/gemm_one.py $1 $2 $3 $5 > $OUT/log.txt 2>&1
