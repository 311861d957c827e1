#!/bin/bash
# Per-rank workloads of N-GPU runs of the 100M index, measured on ONE GPU: 1-D (vector shards, full batch) and the 2-D
# (vector shards x query groups) decomposition of DESIGN.md section 6.  No collective in these numbers.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
TAG=${TAG:-sm}
: > gpurun_out/${TAG}_per_rank_workloads.txt
for cfg in "50000000 1024" "25000000 1024" "12500000 1024" "100000000 512" "50000000 512" "25000000 512" "50000000 256"; do
  set -- $cfg
  timeout 600 python bench.py --n $1 --batch $2 --steps 20 --warmup 5 --cpu-queries 0 --no-recall --no-configs > gpurun_out/${TAG}_n$1_b$2.json 2> gpurun_out/${TAG}_n$1_b$2.log
  python - gpurun_out/${TAG}_n$1_b$2.json $1 $2 >> gpurun_out/${TAG}_per_rank_workloads.txt <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["stage_ms_per_step"]
print(f"n={sys.argv[2]} batch={sys.argv[3]}: {j['ms_per_step']:.3f} ms/step, scan {s['scan']:.3f}, fixed {j['ms_per_step']-s['scan']:.3f}, {j['value']:.0f} q/s of this one rank")
P
done
cat gpurun_out/${TAG}_per_rank_workloads.txt
