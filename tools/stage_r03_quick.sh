#!/bin/bash
# quick check on one GPU box: the GPU suite, then the default bench line (all configs) -> gpurun_out/${TAG}_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03q}
timeout 900 python -m pytest tests -q -x -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 5 gpurun_out/${TAG}_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.log; echo "exit $?" >> gpurun_out/${TAG}_bench.log
tail -n 4 gpurun_out/${TAG}_bench.log | cut -c1-400
