# the headline workload away from its headline point: sampling temperature (acceptance) and table size
set -e
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r01_g_free_variants.jsonl
for th in 10 1 0.1; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --theta $th >> gpurun_out/r01_g_free_variants.jsonl
done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --d-omega 1e-5 >> gpurun_out/r01_g_free_variants.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r01_g_free_variants.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][50:130], "| moves/s %.3g sweeps/s %.3g acc/prop %.3f alg GB/s %.0f" % (d["value"], d["sweeps_per_s"], d["acceptance"]["accepted_moves_per_proposal"], d["roofline"]["achieved"]))
PY
