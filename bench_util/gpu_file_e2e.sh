#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_file_e2e.py -x -q --timeout 400 --timeout-method thread > gpurun_out/r2_file.log 2>&1 || { tail -40 gpurun_out/r2_file.log; exit 1; }
tail -3 gpurun_out/r2_file.log
timeout -k 10 300 python bench.py --workload strings_file --headline-only > gpurun_out/r2_file.json 2>gpurun_out/r2_file.err || { tail -20 gpurun_out/r2_file.err; exit 1; }
cat gpurun_out/r2_file.json
