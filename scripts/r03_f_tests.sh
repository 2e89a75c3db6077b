#!/bin/bash
# round 3, call f: Theta_opt tests (peak, edge) + PEAK_PROF section profile of the two-axis peak kernel
set -o pipefail
mkdir -p gpurun_out
SAC_HIP_LIB=$PWD/tools/ab/libsac_hip_peakprof.so timeout -k 10 300 python tools/bench_param.py peak2 > gpurun_out/r03_peak_prof.txt 2>&1 || exit 1
SAC_HIP_LIB=$PWD/tools/ab/libsac_hip_peakprof.so timeout -k 10 300 python tools/bench_param.py peak >> gpurun_out/r03_peak_prof.txt 2>&1 || exit 1
timeout -k 10 300 python tools/bench_param.py peak2 >> gpurun_out/r03_peak_prof.txt 2>&1 || exit 1
cut -c1-700 gpurun_out/r03_peak_prof.txt
