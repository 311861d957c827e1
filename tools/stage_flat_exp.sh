#!/bin/bash
# Flat GEMM evidence on one GPU box: LDS fill-rate microbenchmark, parity tests, rocprof kernel averages and PMC FETCH_SIZE of
# k_flat_gemm2 with the walking map (default) and the db-stationary passes (RSX_FG2_WALK=0).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-fx}
for i in 2 3 5 6 9 10; do timeout 60 tools/proto/lds_fill_rate $i 2>&1 | tail -1; done > gpurun_out/${TAG}_lds_fill_rate.txt
timeout 900 python -m pytest tests/test_gpu_flat.py tests/test_gpu_scale.py -q -m gpu -x --timeout 600 -p no:cacheprovider -k "flat or Flat" > gpurun_out/${TAG}_pytest_flat.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_flat.log
run() {  # name, env...
  local name=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$name" -o $TAG -- python "$OLDPWD/tools/bench_configs.py" flat --check ${CHECK:-0} --steps 3 > "$OLDPWD/gpurun_out/${TAG}_flat_$name.json" 2> "$OLDPWD/gpurun_out/${TAG}_flat_$name.log" ); echo "exit $?" >> gpurun_out/${TAG}_flat_$name.log
  python tools/rocprof_summary.py gpurun_out/prof_$name/${TAG}_results.db gpurun_out/${TAG}_rocprof_flat_$name.md "Flat 10M x 768 batch 1024 ($name: $*)" > /dev/null 2>&1
  rm -rf gpurun_out/prof_$name
  grep -E "k_flat_gemm2" gpurun_out/${TAG}_rocprof_flat_$name.md | head -3
}
pmc() {  # name, env...
  local name=$1; shift
  ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_$name" -o $TAG -- python "$OLDPWD/tools/bench_configs.py" flat --check 0 --steps 2 > /dev/null 2> "$OLDPWD/gpurun_out/${TAG}_pmc_$name.log" ); echo "exit $?" >> gpurun_out/${TAG}_pmc_$name.log
  echo "## $name: $*" >> gpurun_out/${TAG}_pmc_fetch_flat.md
  python tools/pmc_summary.py gpurun_out/pmc_$name/${TAG}_results.db gpurun_out/${TAG}_pmc_fetch_flat.md '%k_flat_gemm2%'
  rm -rf gpurun_out/pmc_$name
}
rm -f gpurun_out/${TAG}_pmc_fetch_flat.md
CHECK=64 run walk RSX_FG2_WALK=1
CHECK=64 run stationary RSX_FG2_WALK=0
if [ -n "${PMC:-}" ]; then
pmc walk RSX_FG2_WALK=1
pmc stationary RSX_FG2_WALK=0
fi
cat gpurun_out/${TAG}_pmc_fetch_flat.md gpurun_out/${TAG}_lds_fill_rate.txt 2>/dev/null; tail -3 gpurun_out/${TAG}_pytest_flat.log
