#!/bin/bash
# last call of the round: GPU suite + PMC FETCH_SIZE pass for the stamped traffic file
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03zz}
timeout 300 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 2 gpurun_out/${TAG}_pytest_gpu.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs --no-faiss > "$OLDPWD/gpurun_out/${TAG}_pmc_bench.json" 2> "$OLDPWD/gpurun_out/${TAG}_pmc.log" ); echo "exit $?" >> gpurun_out/${TAG}_pmc.log
python tools/pmc_summary.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/${TAG}_pmc_fetch_size.md '%k_pq_scan%' '%k_pq_prepass%' > /dev/null
python tools/update_pmc_traffic.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/pmc_traffic.json | cut -c1-300
rm -rf gpurun_out/pmc_fetch
