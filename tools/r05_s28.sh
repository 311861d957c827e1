#!/bin/bash
# Round-5 GPU session 28 (the round's last GPU seconds): k_pq_scan_rot at M = 96 with two code blocks in flight per wave — re-stamp FETCH_SIZE / MFMA busy, then the IVF-PQ parity tests
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export TAG=${TAG:-r05ze}
bash tools/gpu_round.sh pmc_fetch
grep -E "hbm_bytes|source_sha|mfma_busy_frac|avg_us" $O/pmc_traffic.json
timeout 100 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_distributions.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4 > $O/${TAG}_tests.txt
cat $O/${TAG}_tests.txt
