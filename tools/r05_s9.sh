#!/bin/bash
# Round-5 GPU session 9: long-lists-first item order — GPU suite, interleaved A/B on the headline, hot probe sets, N = 8 rank shard
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05i}
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
tail -n 12 $O/${T}_pytest_gpu.txt | cut -c1-250
: > $O/${T}_ab.txt
for p in "scan_order=1" "scan_order=0" "scan_order=1" "scan_order=0" "scan_order=1"; do
  timeout 600 python bench.py --steps 20 --warmup 5 $FAST --param $p > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== $p" >> $O/${T}_ab.txt; python tools/show_bench.py $O/${T}_ab_tmp.json >> $O/${T}_ab.txt
done
for p in "scan_order=1" "scan_order=0"; do
  timeout 600 python bench.py --n 12500000 --steps 20 --warmup 5 $FAST --param $p > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== 12.5M $p" >> $O/${T}_ab.txt; python tools/show_bench.py $O/${T}_ab_tmp.json "n=12.5M" >> $O/${T}_ab.txt
done
cat $O/${T}_ab.txt | cut -c1-330
for p in "scan_order=1" "scan_order=0"; do
  timeout 900 python tools/bench_dist.py hot --check 0 --param $p > $O/${T}_hot_$p.jsonl 2> $O/${T}_hot.log
  python - <<PY
import json
for l in open("$O/${T}_hot_$p.jsonl"):
    r=json.loads(l)
    for k,v in r.items(): print("$p", k, v["ms_per_step"], "scan", v["stage_ms"]["scan"], "fb", v["exact_fallback_queries_per_step"])
PY
done
