#!/bin/bash
# IVF-Flat check: GPU suite, then the reference's IVF-Flat operating point (20M, nlist 2048, nprobe 128) and the 100M config -> gpurun_out/${TAG}_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03ivf}
timeout 900 python -m pytest tests -q -x -m gpu --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt
timeout 600 python tools/bench_configs.py ivfflat --n 20000000 --nlist 2048 --nprobe 128 --check 2 --steps 3 > gpurun_out/${TAG}_ivfflat20M.json 2> gpurun_out/${TAG}_ivfflat20M.log; echo "exit $?" >> gpurun_out/${TAG}_ivfflat20M.log
cut -c1-900 gpurun_out/${TAG}_ivfflat20M.json; tail -n 2 gpurun_out/${TAG}_ivfflat20M.log
if [ -n "$ALSO100M" ]; then
timeout 900 python tools/bench_configs.py ivfflat --n 100000000 --nlist 4096 --nprobe 32 --check 2 --steps 3 > gpurun_out/${TAG}_ivfflat100M.json 2> gpurun_out/${TAG}_ivfflat100M.log; echo "exit $?" >> gpurun_out/${TAG}_ivfflat100M.log
cut -c1-700 gpurun_out/${TAG}_ivfflat100M.json
fi
