#!/bin/bash
# Round-5 GPU session 10: A/B of the item order after the pair-scan tune-up (100M and the 12.5M rank shard), parallel-entry finalize for batches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05j}
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
: > $O/${T}_ab.txt
for n in 100000000 12500000; do
for p in "scan_order=1" "scan_order=0" "scan_order=1" "scan_order=0"; do
  timeout 600 python bench.py --n $n --steps 20 --warmup 5 $FAST --param $p > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== n=$n $p" >> $O/${T}_ab.txt; python tools/show_bench.py $O/${T}_ab_tmp.json "n=$n" >> $O/${T}_ab.txt
done
done
for v in 0 1 0 1; do
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_FIN_PAR=$v timeout 600 python bench.py --steps 20 --warmup 5 $FAST > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== measure lib RSX_FIN_PAR=$v" >> $O/${T}_ab.txt; python tools/show_bench.py $O/${T}_ab_tmp.json >> $O/${T}_ab.txt
done
cat $O/${T}_ab.txt | cut -c1-330
timeout 900 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider -k "latency or golden or ivf" > $O/${T}_pytest_subset.txt 2>&1; tail -n 4 $O/${T}_pytest_subset.txt | cut -c1-200
