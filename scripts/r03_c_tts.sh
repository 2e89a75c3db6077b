#!/bin/bash
# round 3: the no-alloc test again, then the complete reference runs (time to solution) under the chain-mean chi2_min
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_zz_no_alloc.py tests/test_gpu_comm_abi.py > gpurun_out/r03_c_tests.log 2>&1 || { tail -40 gpurun_out/r03_c_tests.log; exit 1; }
tail -3 gpurun_out/r03_c_tests.log
timeout -k 10 300 python tools/time_to_solution.py free 64 > gpurun_out/r03_tts_free.jsonl 2> gpurun_out/r03_tts_free.err
cat gpurun_out/r03_tts_free.jsonl
timeout -k 10 700 python tools/time_to_solution.py edge 64 nospread 1e-4 > gpurun_out/r03_tts_edge.jsonl 2> gpurun_out/r03_tts_edge.err
cat gpurun_out/r03_tts_edge.jsonl
