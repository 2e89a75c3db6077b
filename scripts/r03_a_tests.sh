#!/bin/bash
# round 3, first GPU pass: the merge behind the C ABI + the drivers that now ride on it
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest -x -v --timeout 300 --timeout-method thread tests/test_gpu_comm_abi.py tests/test_gpu_rccl_one_rank.py \
   tests/test_gpu_dist_drivers.py tests/test_gpu_cross_val.py tests/test_gpu_driver.py tests/test_gpu_peak_driver.py tests/test_gpu_edge_driver.py \
   > gpurun_out/r03_a_tests.log 2>&1 || { tail -60 gpurun_out/r03_a_tests.log; exit 1; }
tail -40 gpurun_out/r03_a_tests.log
