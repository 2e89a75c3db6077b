#!/bin/bash
# round 3, call g: peak parity after the parked loads + throughput of the peak kernels + section profile
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_peak.py tests/test_gpu_free.py tests/test_gpu_replay.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r03_g_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03_g_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bench_param.py peak2 > gpurun_out/r03_g_peak.txt 2>&1 || exit 1
timeout -k 10 300 python tools/bench_param.py peak >> gpurun_out/r03_g_peak.txt 2>&1 || exit 1
SAC_HIP_LIB=$PWD/tools/ab/libsac_hip_peakprof.so timeout -k 10 300 python tools/bench_param.py peak2 2>&1 | grep "sweeps=100\|workload" >> gpurun_out/r03_g_peak.txt || exit 1
timeout -k 10 300 python tools/peak_latency.py 64 >> gpurun_out/r03_g_peak.txt 2>&1 || exit 1
cut -c1-600 gpurun_out/r03_g_peak.txt
