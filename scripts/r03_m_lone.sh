#!/bin/bash
# round 3, call m: section profile of a lone peak chain (fermionic2 input, 64 chains on the card)
set -o pipefail
mkdir -p gpurun_out
SAC_HIP_LIB=$PWD/tools/ab/libsac_hip_peakprof.so timeout -k 10 300 python tools/peak_latency.py 64 2>&1 | grep "peak prof\|chains" | tail -8 | cut -c1-500 > gpurun_out/r03_m_lone.txt
cat gpurun_out/r03_m_lone.txt
