#!/bin/bash
# round 3, call e: the Theta_opt chain-count tests, configs[0], the new parity case
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests/test_gpu_configs0.py tests/test_gpu_free.py tests/test_gpu_theta_opt.py -x -v -s --timeout 600 --timeout-method thread -m gpu > gpurun_out/r03_e_tests.log 2>&1
rc=$?
tail -40 gpurun_out/r03_e_tests.log
exit $rc
