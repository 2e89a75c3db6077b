# additional round-2 counter passes: the peak entry of bench.py's hbm_regime
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r02
mkdir -p $O
run() {
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 170 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $O/$name.log; exit 1; }
  echo "$name done"
}
run pkhb_p3 "FETCH_SIZE GRBM_GUI_ACTIVE" $R/tools/bench_param.py peak_hbm && run pkhb_p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $R/tools/bench_param.py peak_hbm
run pkh4_p3 "FETCH_SIZE GRBM_GUI_ACTIVE" $R/tools/bench_param.py peak_hbm --d-omega 2.5e-6 && run pkh4_p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $R/tools/bench_param.py peak_hbm --d-omega 2.5e-6
find $O -name "*agent_info.csv" -delete
cd $R && timeout -k 10 120 python tools/bench_param.py peak_hbm && timeout -k 10 120 python tools/bench_param.py peak_hbm --d-omega 2.5e-6
