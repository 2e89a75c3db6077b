#!/bin/bash
# round 3, call n: hypothesis pairs (free_sweep_hyp.h): parity against the oracle, then the latency of one chain in the three step modes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_free_hyp.py -x -q --timeout 280 --timeout-method thread -m gpu > gpurun_out/r03_n_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r03_n_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/hyp_scan.py 1 64 > gpurun_out/r03_n_hyp_scan.jsonl 2>&1
rc=$?
cat gpurun_out/r03_n_hyp_scan.jsonl | cut -c1-400
exit $rc
