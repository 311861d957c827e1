#!/bin/bash
# Trimmed evidence pass (end of round 2, after the fixed-cost work): GPU tests, rocprofv3 stats, PMC FETCH_SIZE (stamped
# traffic file), one SQ counter pass, the full default bench line, single-query latency.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r02c}
BARGS="--steps 5 --warmup 2 --cpu-queries 0 --no-recall --no-configs"
timeout 120 python -m pytest tests -q -m gpu --timeout 100 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o $TAG -- python "$OLDPWD/bench.py" $BARGS > /dev/null 2> "$OLDPWD/gpurun_out/prof.log" )
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py $BARGS)"
rm -rf gpurun_out/prof
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > /dev/null 2> "$OLDPWD/gpurun_out/pmc.log" )
rm -f gpurun_out/${TAG}_pmc_fetch_size.md
python tools/pmc_summary.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/${TAG}_pmc_fetch_size.md '%k_pq_scan%' '%k_pq_prepass%' '%k_pq_rot%'
python tools/update_pmc_traffic.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch
timeout 240 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_ivfpq100M.json 2> gpurun_out/${TAG}_bench.log; echo "exit $?" >> gpurun_out/${TAG}_bench.log
timeout 60 python tools/bench_configs.py latency > gpurun_out/${TAG}_latency_ivfpq100M.json 2> gpurun_out/${TAG}_latency.log
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d "$OLDPWD/gpurun_out/pmc_sq" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > /dev/null 2> "$OLDPWD/gpurun_out/pmc_sq.log" )
rm -f gpurun_out/${TAG}_pmc_sq_counters.md
python tools/pmc_summary.py gpurun_out/pmc_sq/${TAG}_results.db gpurun_out/${TAG}_pmc_sq_counters.md '%k_pq_scan%' '%k_pq_rot%' '%k_finalize%' '%k_pq_lut%' '%k_pq_prepass%'
rm -rf gpurun_out/pmc_sq
tail -2 gpurun_out/${TAG}_pytest_gpu.txt; tail -2 gpurun_out/${TAG}_bench.log | cut -c1-200; cat gpurun_out/pmc_traffic.json | head -12
