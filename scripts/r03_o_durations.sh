#!/bin/bash
# round 3, call o: the whole GPU suite with per-test durations (the driver's round-end step allows pytest 900 s)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -x -q --timeout 600 --timeout-method thread -m gpu --durations=70 > gpurun_out/r03_o_tests.log 2>&1
rc=$?
tail -90 gpurun_out/r03_o_tests.log | cut -c1-200
exit $rc
