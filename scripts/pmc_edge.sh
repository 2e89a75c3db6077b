set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_edge1 -o p1 -- python3 $R/tools/bench_param.py edge > $R/gpurun_out/pmc_edge1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_edge2 -o p2 -- python3 $R/tools/bench_param.py edge > $R/gpurun_out/pmc_edge2.log 2>&1
find $R/gpurun_out/pmc_edge1 $R/gpurun_out/pmc_edge2 -name "*.csv" | head
