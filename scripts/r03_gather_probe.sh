#!/bin/bash
# Round 3: gather ceilings on the sampler's own column distribution -- how they move with the rows in flight, the load
# width and the wave count (configs[1]'s 48.8 MiB table), and the same probe on the HBM-resident tables (488 / 1953 MiB)
# plus a deliberately flat support.  Output: gpurun_out/r03_gather_probe.txt
set -e
out=gpurun_out/r03_gather_probe.txt
mkdir -p gpurun_out
: > $out
timeout -k 10 300 python tools/dump_positions.py gpurun_out/pos_1e-4.bin 4096 1e-4 >> $out
for d in 8 16 32; do for w in x2 x4; do
  timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 $d $w >> $out
done; done
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 16 x2 8192 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 16 x4 8192 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 16 x2 2048 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 16 x4 2048 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-4.bin 100000 16 x2 4096 flat >> $out
timeout -k 10 300 python tools/dump_positions.py gpurun_out/pos_1e-5.bin 4096 1e-5 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-5.bin 100000 16 x2 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-5.bin 100000 32 x2 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_1e-5.bin 100000 16 x2 4096 flat >> $out
timeout -k 10 300 python tools/dump_positions.py gpurun_out/pos_2.5e-6.bin 4096 2.5e-6 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_2.5e-6.bin 100000 16 x2 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_2.5e-6.bin 100000 32 x2 >> $out
timeout -k 10 60 tools/gather_probe --positions gpurun_out/pos_2.5e-6.bin 100000 16 x2 4096 flat >> $out
rm -f gpurun_out/pos_*.bin
cat $out
