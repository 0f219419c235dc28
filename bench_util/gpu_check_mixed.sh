#!/bin/bash
# parity subset, then the mixed_short_runs bench line; stops at the first failure
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_strings.py tests/test_gpu_synth.py tests/test_gpu_decode_names.py tests/test_gpu_no_entries.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r2_t.log 2>&1 || { tail -30 gpurun_out/r2_t.log; exit 1; }
tail -3 gpurun_out/r2_t.log
timeout -k 10 200 python bench.py --workload mixed_short_runs --headline-only > gpurun_out/r2_mixed.json 2>gpurun_out/r2_mixed.err
cat gpurun_out/r2_mixed.json
bash bench_util/gpu_bench_one.sh mixed_short_runs
bash bench_util/gpu_bench_one.sh delta_w10
