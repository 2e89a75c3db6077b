#!/bin/bash
# round 3, call k: final bench line + the same under rocprofv3 --kernel-trace --stats + the peak 4M-column gather ceiling
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r03_gather_probe_peak4m.txt
: > $out
timeout -k 10 300 python tools/dump_positions_peak.py gpurun_out/pos_peak4.bin 4096 one 2.5e-6 >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak4.bin 100000 16 x2 >> $out || exit 1
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_peak4.bin 100000 32 x2 >> $out || exit 1
rm -f gpurun_out/pos_peak4.bin
cat $out
bash scripts/r02_bench_profile.sh r03_k
