#!/bin/bash
# GPU parity suite with the margin table on: every assert_close() records how much of its bound it used
# (tests/conftest.py); the table goes to gpurun_out/parity_margin.json -> copy to profiles/r03_parity_margin.json
export HD_PARITY_MARGIN_OUT=$GRAFT_REPO_ROOT/gpurun_out/parity_margin.json
cd $GRAFT_REPO_ROOT && timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu -s --timeout 180 --timeout-method thread > gpurun_out/parity_margin.log 2>&1
rc=$?
tail -5 gpurun_out/parity_margin.log
exit $rc
