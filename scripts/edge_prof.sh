# section profile (chain 0) of EDGE_PROF builds of the library
cd $GRAFT_REPO_ROOT
for lib in $LIBS; do
  echo "== $lib"
  SAC_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python tools/bench_param.py edge 2>&1 | grep -E "^prof|chain_sweeps" | tail -4 | cut -c1-400
done
