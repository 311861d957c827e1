#!/bin/bash
# Flat check: GPU suite, then Flat 10M IP / L2 with the oracle spot check, rocprof kernel stats -> gpurun_out/${TAG}_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03flat}
timeout 900 python -m pytest tests -q -x -m gpu --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_flat" -o $TAG -- python "$OLDPWD/tools/bench_configs.py" flat --check 64 --steps 3 > "$OLDPWD/gpurun_out/${TAG}_flat10M.json" 2> "$OLDPWD/gpurun_out/${TAG}_flat10M.log" ); echo "exit $?" >> gpurun_out/${TAG}_flat10M.log
python tools/rocprof_summary.py gpurun_out/prof_flat/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_flat10M.md "Flat 10M x 768 batch 1024 (tools/bench_configs.py flat --check 64 --steps 3)" > /dev/null
rm -rf gpurun_out/prof_flat
cat gpurun_out/${TAG}_flat10M.json | cut -c1-700; grep -E "flat_gemm|k_select|k_finalize" gpurun_out/${TAG}_rocprof_stats_flat10M.md | head -8 | cut -c1-110
