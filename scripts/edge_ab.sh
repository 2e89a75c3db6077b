# edge kernel A/B: parity tests on the new library, then the edge throughput line on the baseline build and the new one
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_edge.py -x -q --timeout 120 --timeout-method thread > gpurun_out/edge_ab_tests.txt 2>&1 || { tail -30 gpurun_out/edge_ab_tests.txt; exit 1; }
tail -3 gpurun_out/edge_ab_tests.txt
for lib in variants/libsac_base.so sac_amd/libsac_hip.so ${EXTRA_LIBS}; do
  echo "== $lib" >> gpurun_out/edge_ab.txt
  SAC_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python tools/bench_param.py edge >> gpurun_out/edge_ab.txt 2>&1
  SAC_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python tools/bench_param.py edge >> gpurun_out/edge_ab.txt 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/edge_ab.txt'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); print({k: d[k] for k in ('chain_sweeps_per_s','lookups_per_s','stage_ms','table_GBps','frac')})
PY
