set -e
cd $GRAFT_REPO_ROOT
for spl in 25 50 100 200 400; do
  timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu --sweeps-per-launch $spl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($spl, round(d['value']/1e9,4), 'e9 moves/s', round(d['ms_per_step'],1), 'ms/step', round(d['roofline']['avg_launch_ms'],2), 'ms/launch')" | tee -a gpurun_out/spl_scan.txt
done
