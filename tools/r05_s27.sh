#!/bin/bash
# Round-5 GPU session 27: code blocks in flight per wave of k_pq_scan_rot (ROT_DEPTH 4 = shipped, 2, 1; variant libraries built beside the product's), headline only
# (historical: librsx_depth1/2.so were variant builds of k_pq_rot.hip with -DROT_DEPTH=1/2; depth 2 at M = 96 became the product's setting)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05zd}
: > $O/${T}_rot_depth.txt
for lib in librsx.so librsx_depth2.so librsx_depth1.so librsx.so librsx_depth2.so; do
  RSX_LIB=$R/retrieval-scaling_amd/csrc/$lib timeout 200 python bench.py --steps 20 --warmup 5 --cpu-queries 0 --no-recall --no-configs > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  python tools/show_bench.py $O/${T}_tmp.json "$lib" | head -1 >> $O/${T}_rot_depth.txt
done
cat $O/${T}_rot_depth.txt | cut -c1-260
tail -2 $O/${T}_tmp.log
