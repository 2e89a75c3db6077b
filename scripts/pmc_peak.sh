set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_peak1 -o p1 -- python3 $R/tools/bench_param.py peak > $R/gpurun_out/pmc_peak1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_peak2 -o p2 -- python3 $R/tools/bench_param.py peak > $R/gpurun_out/pmc_peak2.log 2>&1
ls $R/gpurun_out/pmc_peak1 $R/gpurun_out/pmc_peak2
