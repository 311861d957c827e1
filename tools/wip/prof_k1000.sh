#!/bin/bash
# rocprof kernel stats of Flat / IVF-Flat at the reference's n_docs (k = 1000): gpurun_out/<tag>_rocprof_stats_<which>_k1000.md
# usage: TAG=.. WHICH="flat ivfflat ivfflat128" tools/wip/prof_k1000.sh
TAG=${TAG:-k1000}; ROOT=$PWD; O=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in ${WHICH:-flat ivfflat ivfflat128}; do
  case $w in
    flat) ARGS="flat --k 1000 --check 0 --steps 3";;
    ivfflat) ARGS="ivfflat --k 1000 --check 0 --steps 3";;
    ivfflat128) ARGS="ivfflat --nlist 2048 --nprobe 128 --k 1000 --check 0 --steps 3";;
  esac
  rm -rf /tmp/pk
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pk -o pk -- python $ROOT/tools/bench_configs.py $ARGS > $O/${TAG}_${w}_k1000.json 2> /tmp/pk.log
  python $ROOT/tools/rocprof_summary.py /tmp/pk/pk_results.db $O/${TAG}_rocprof_stats_${w}_k1000.md "tools/bench_configs.py $ARGS"
  cut -c1-300 $O/${TAG}_${w}_k1000.json
done
