#!/bin/bash
# the whole GPU suite, one process, output to gpurun_out/r2_full.log
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_full.log 2>&1
rc=$?
tail -5 gpurun_out/r2_full.log
exit $rc
