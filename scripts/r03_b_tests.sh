#!/bin/bash
# round 3, second GPU pass: driver tests under the chain-mean chi2_min, then the complete reference runs (time to solution)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -x -v --timeout 300 --timeout-method thread tests/test_gpu_driver.py tests/test_gpu_peak_driver.py \
   tests/test_gpu_edge_driver.py tests/test_gpu_replay.py tests/test_zz_no_alloc.py > gpurun_out/r03_b_tests.log 2>&1 || { tail -60 gpurun_out/r03_b_tests.log; exit 1; }
tail -25 gpurun_out/r03_b_tests.log
timeout -k 10 300 python tools/time_to_solution.py free 64 > gpurun_out/r03_tts_free.jsonl 2> gpurun_out/r03_tts_free.err
cat gpurun_out/r03_tts_free.jsonl
timeout -k 10 700 python tools/time_to_solution.py edge 64 nospread 1e-4 > gpurun_out/r03_tts_edge.jsonl 2> gpurun_out/r03_tts_edge.err
cat gpurun_out/r03_tts_edge.jsonl
