# additional round-2 counter passes: sac/peak with both frequency axes (the weight-transfer kernels, XFER = true)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r02
mkdir -p $O
run() {
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 170 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $O/$name.log; exit 1; }
  echo "$name done"
}
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
run pk2_p1 "$P1" $R/tools/bench_param.py peak2 && run pk2_p3 "FETCH_SIZE GRBM_GUI_ACTIVE" $R/tools/bench_param.py peak2 && \
run pk2_p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $R/tools/bench_param.py peak2 && run pk2_p2 "$P2" $R/tools/bench_param.py peak2
find $O -name "*agent_info.csv" -delete
cd $R && timeout -k 10 120 python tools/bench_param.py peak2
