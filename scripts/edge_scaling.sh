set -e
cd $GRAFT_REPO_ROOT
for c in 256 1024 2048 4096; do
  for d in 1e-5 1e-3; do
    timeout -k 10 120 python tools/bench_param.py edge --chains $c --d-omega $d | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($c, '$d', round(d['chain_sweeps_per_s']), round(d['lookups_per_s']/1e6), 'Mlookups/s', round(d['stage_ms'],1),'ms')" | tee -a gpurun_out/edge_scaling.txt
  done
done
