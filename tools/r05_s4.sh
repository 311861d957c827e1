#!/bin/bash
# Round-5 GPU session 4: persistent LUT build (tests + timing), coarse-GEMM spread experiment, distribution legs, new tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r05d
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
timeout 1200 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider -k "ivf or golden or dist or lut or shape" > $O/${T}_pytest_subset.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_subset.txt
tail -n 8 $O/${T}_pytest_subset.txt | cut -c1-250
: > $O/${T}_ab.txt
run() { # label, env..., -- args
  local label=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
for v in "lib" "measure_spread0" "measure_spread64" "measure_spread100" "lib2"; do
  case $v in
    lib|lib2) ENVV="" ;;
    measure_spread0) ENVV="RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so" ;;
    measure_spread64) ENVV="RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_COARSE_SPREAD=64" ;;
    measure_spread100) ENVV="RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_COARSE_SPREAD=100" ;;
  esac
  env $ENVV timeout 600 python bench.py --steps 20 --warmup 5 $FAST > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== $v" >> $O/${T}_ab.txt
  python tools/show_bench.py $O/${T}_ab_tmp.json >> $O/${T}_ab.txt
done
cat $O/${T}_ab.txt | cut -c1-330
# kernel-level view of the fixed part (rocprof stats of the headline loop)
( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d /tmp/prof -o $T -- python $R/bench.py --steps 5 --warmup 2 $FAST > /dev/null 2> $O/${T}_prof.log ); echo "exit $?" >> $O/${T}_prof.log
python tools/rocprof_summary.py /tmp/prof/${T}_results.db $O/${T}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py --steps 5 --warmup 2 $FAST)"
head -n 40 $O/${T}_rocprof_stats_ivfpq100M.md | cut -c1-200
# distribution legs
timeout 1500 python tools/bench_dist.py hot informative norm_skew > $O/${T}_dist.jsonl 2> $O/${T}_dist.log; echo "exit $?" >> $O/${T}_dist.log
cut -c1-1800 $O/${T}_dist.jsonl; tail -n 5 $O/${T}_dist.log | cut -c1-300
