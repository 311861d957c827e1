#!/bin/bash
# vector-shard per-rank workloads (full batch) of 2 / 4 / 8-GPU runs of the 100M index, measured on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
TAG=${TAG:-r02c}
: > gpurun_out/${TAG}_per_rank_workloads.txt
for n in 50000000 25000000 12500000; do
  timeout 60 python bench.py --n $n --steps 20 --warmup 5 --cpu-queries 0 --no-recall --no-configs > gpurun_out/pr_$n.json 2>/dev/null
  python - gpurun_out/pr_$n.json $n >> gpurun_out/${TAG}_per_rank_workloads.txt <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["stage_ms_per_step"]
print("n=%s batch=1024: %.3f ms/step, scan %.3f, fixed %.3f, %.0f q/s of this one rank" % (sys.argv[2], j["ms_per_step"], s["scan"], j["ms_per_step"]-s["scan"], j["value"]))
P
done
cat gpurun_out/${TAG}_per_rank_workloads.txt
