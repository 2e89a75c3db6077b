# round-3 rocprofv3 counter passes (one --pmc set per run, each its own process under a hard time limit); the summary
# goes to profiles/ through tools/pmc_report.py by hand
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r03
mkdir -p $O
# the build these passes measure (bench.py quotes profiles/traffic.json only for this build)
(cd $R && python3 -c "from sac_amd.lib import build_id; print(build_id())") > $O/build_id.txt
run() {   # name, counters, program args...
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 170 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $O/$name.log; exit 1; }
  echo "$name done"
}
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
FREE="$R/bench.py --steps 3 --warmup 1 --no-cpu --no-hbm-regime"
EDGE="$R/tools/bench_param.py edge"
PEAK="$R/tools/bench_param.py peak"
PEAK2="$R/tools/bench_param.py peak2"
PKHB="$R/tools/bench_param.py peak_hbm --d-omega 1e-5"
PKH4="$R/tools/bench_param.py peak_hbm --d-omega 2.5e-6"
HBM1="$R/tools/bench_param.py free_hbm --d-omega 1e-5"
HBM4="$R/tools/bench_param.py free_hbm --d-omega 2.5e-6"
PACK="$R/tools/bench_param.py free_packed"
PART=${1:-all}
if [ "$PART" = "a" ] || [ "$PART" = "all" ]; then
run free_p1 "$P1" $FREE && run free_p3 "$P3" $FREE && run free_p4 "$P4" $FREE && run free_p2 "$P2" $FREE && \
run hbm1_p3 "$P3" $HBM1 && run hbm1_p4 "$P4" $HBM1 && run hbm4_p3 "$P3" $HBM4 && run hbm4_p4 "$P4" $HBM4 && \
run pk2_p1 "$P1" $PEAK2 && run pk2_p3 "$P3" $PEAK2 && run pk2_p4 "$P4" $PEAK2 && \
run pkhb_p3 "$P3" $PKHB && run pkhb_p4 "$P4" $PKHB && run pkh4_p3 "$P3" $PKH4 && run pkh4_p4 "$P4" $PKH4 || exit 1
fi
if [ "$PART" = "b" ] || [ "$PART" = "all" ]; then
run peak_p1 "$P1" $PEAK && run peak_p3 "$P3" $PEAK && run peak_p4 "$P4" $PEAK && run peak_p2 "$P2" $PEAK && \
run edge_p1 "$P1" $EDGE && run edge_p3 "$P3" $EDGE && run edge_p4 "$P4" $EDGE && run edge_p2 "$P2" $EDGE && \
run pack_p1 "$P1" $PACK && run pack_p3 "$P3" $PACK && run pack_p4 "$P4" $PACK && run pack_p2 "$P2" $PACK || exit 1
fi
# keep the pulled set small: only the counter tables
find $O -name "*agent_info.csv" -delete
du -sh $O
