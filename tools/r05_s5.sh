#!/bin/bash
# Round-5 GPU session 5: full GPU suite (KPv finalize, distribution tests), phase traces of finalize / gather-select, Flat + IVF-Flat k = 1000, distribution legs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r05e
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
tail -n 12 $O/${T}_pytest_gpu.txt | cut -c1-250
RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so timeout 600 python tools/exp_ft_trace.py 100000000 10 100 1000 > $O/${T}_ft_trace.txt 2> $O/${T}_ft_trace.log; echo "exit $?" >> $O/${T}_ft_trace.log
cat $O/${T}_ft_trace.txt | cut -c1-400; tail -n 3 $O/${T}_ft_trace.log | cut -c1-300
timeout 900 python tools/bench_configs.py flat --check 4 --steps 3 --ks 1000 > $O/${T}_flat10M.json 2> $O/${T}_flat10M.log; echo "exit $?" >> $O/${T}_flat10M.log
cut -c1-1500 $O/${T}_flat10M.json
timeout 900 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 --ks 1000 > $O/${T}_ivfflat20M.json 2> $O/${T}_ivfflat20M.log; echo "exit $?" >> $O/${T}_ivfflat20M.log
cut -c1-1500 $O/${T}_ivfflat20M.json
timeout 1500 python tools/bench_dist.py hot informative > $O/${T}_dist.jsonl 2> $O/${T}_dist.log; echo "exit $?" >> $O/${T}_dist.log
cut -c1-1500 $O/${T}_dist.jsonl; tail -n 3 $O/${T}_dist.log | cut -c1-300
