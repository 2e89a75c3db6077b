#!/bin/bash
# round 3, call l: the RCCL paths after the library-selection change: ABI merge tests (no torch), torch-driven one-rank tests incl.
# bench.py forced to its N>1 code path, the dist driver tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_comm_abi.py tests/test_gpu_rccl_one_rank.py tests/test_gpu_dist_drivers.py -x -q --timeout 280 --timeout-method thread -m gpu > gpurun_out/r03_l_tests.log 2>&1
rc=$?
tail -30 gpurun_out/r03_l_tests.log | cut -c1-300
exit $rc
