set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > gpurun_out/r01_g_bench.json 2> gpurun_out/r01_g_bench.err
cat gpurun_out/r01_g_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_g -o r01g -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r01_g_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r01_g_rocprof.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_g
