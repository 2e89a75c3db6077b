#!/bin/bash
# round 3, call i: A/B of the single-move ring depth of the peak kernels (8 shipped, 12, 16) with the deeper Gbar refresh batch
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r03_i_peak_ab.txt
: > $out
timeout -k 10 300 python -u -m pytest tests/test_gpu_peak.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -2 >> $out || exit 1
for lib in sac_amd/libsac_hip.so tools/ab/libsac_hip_d12.so tools/ab/libsac_hip_d16.so; do
  echo "== $lib" >> $out
  SAC_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_param.py peak2 2>&1 | cut -c1-330 >> $out || exit 1
  SAC_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_param.py peak 2>&1 | cut -c1-330 >> $out || exit 1
  SAC_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/peak_latency.py 64 >> $out 2>&1 || exit 1
done
cat $out
