#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03m16}
timeout 600 python -m pytest tests -q -x -m gpu --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 12 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-200
if [ -z "$NOBENCH" ]; then
timeout 600 python tools/bench_configs.py ivfpq_ref --steps 3 > gpurun_out/${TAG}_ivfpq_m16.json 2> gpurun_out/${TAG}_ivfpq_m16.log; echo "exit $?" >> gpurun_out/${TAG}_ivfpq_m16.log
cut -c1-1500 gpurun_out/${TAG}_ivfpq_m16.json; tail -n 3 gpurun_out/${TAG}_ivfpq_m16.log | cut -c1-300
fi
