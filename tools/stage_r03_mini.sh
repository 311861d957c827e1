#!/bin/bash
# smallest useful check: GPU suite, then the bench loop alone (no configs / cpu / recall) -> gpurun_out/${TAG}_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03m}
timeout 900 python -m pytest tests -q -x -m gpu --timeout 600 -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -n 3 gpurun_out/${TAG}_pytest_gpu.txt
BARGS="--steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs --no-faiss"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o $TAG -- python "$OLDPWD/bench.py" $BARGS > "$OLDPWD/gpurun_out/${TAG}_prof_bench.json" 2> "$OLDPWD/gpurun_out/${TAG}_prof.log" ); echo "exit $?" >> gpurun_out/${TAG}_prof.log
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py $BARGS)" > /dev/null
rm -rf gpurun_out/prof
python - gpurun_out/${TAG}_prof_bench.json <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["stage_ms_per_step"]
print(j["value"], j["ms_per_step"], json.dumps(s), "fixed", round(j["ms_per_step"]-s["scan"],4))
P
grep -E "k_pq_prepass|k_gemm_exact_n64|gemm_exact<false, false>|lut_tiled|k_finalize|rot_compact|k_select|rot_items|k_pair|k_probe|k_pq_qparam|k_zero|scan_rot|gather" gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md | cut -c1-100
