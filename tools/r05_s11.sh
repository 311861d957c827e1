#!/bin/bash
# Round-5 GPU session 11: table rows non-temporal or not, by shard size (one rank of N = 2 / 4 / 8) — where is the crossover
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05k}
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
: > $O/${T}_ab.txt
for n in 12500000 25000000 50000000 100000000; do
for p in 16520 16544 16520 16544; do
  timeout 600 python bench.py --n $n --steps 20 --warmup 5 $FAST --param pq_pace=$p > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== n=$n pq_pace=$p ($([ $p = 16544 ] && echo nt || echo plain))" >> $O/${T}_ab.txt; python tools/show_bench.py $O/${T}_ab_tmp.json "n=$n" | head -1 >> $O/${T}_ab.txt
done
done
cat $O/${T}_ab.txt | cut -c1-250
