#!/bin/bash
# round 3, call h: peak parity incl. the tiny-state cases, peak drivers, gather ceilings on the peak column distributions
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_peak.py tests/test_gpu_peak_driver.py tests/test_gpu_comm_abi.py tests/test_gpu_replay.py tests/test_gpu_cv_fused.py -x -q --timeout 600 --timeout-method thread -m gpu > gpurun_out/r03_h_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r03_h_tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
out=gpurun_out/r03_gather_probe_peak.txt
: > $out
timeout -k 10 300 python tools/dump_positions_peak.py gpurun_out/pos_peak1.bin 4096 one >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak1.bin 100000 16 x2 >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak1.bin 100000 32 x2 >> $out || exit 1
timeout -k 10 300 python tools/dump_positions_peak.py gpurun_out/pos_peak2.bin 4096 two >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak2.bin 100000 16 x2 >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak2.bin 100000 32 x2 >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak2.bin 100000 16 x2 4096 flat >> $out || exit 1
rm -f gpurun_out/pos_peak*.bin
cat $out
