#!/bin/bash
# round 3: edge second anneal by median first passage -- tests, then the complete reference run
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_comm_abi.py tests/test_gpu_edge_driver.py tests/test_gpu_edge.py > gpurun_out/r03_d_tests.log 2>&1 || { tail -40 gpurun_out/r03_d_tests.log; exit 1; }
tail -3 gpurun_out/r03_d_tests.log
timeout -k 10 800 python tools/time_to_solution.py edge 64 nospread 1e-4 > gpurun_out/r03_tts_edge2.jsonl 2> gpurun_out/r03_tts_edge2.err
cat gpurun_out/r03_tts_edge2.jsonl
