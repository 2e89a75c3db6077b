# round-2 evidence: the default bench line, the same bench under rocprofv3 --kernel-trace --stats (average launch duration of
# the dominant kernel must agree with the live HIP-event figure), summaries copied to profiles/ by hand
set -e
TAG=${1:-r02_d}
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-600 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o ${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-hbm-regime > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.err
find $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv \;
head -5 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
