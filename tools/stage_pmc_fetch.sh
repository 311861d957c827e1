# rocprofv3 PMC pass (FETCH_SIZE) over the bench workload -> gpurun_out/pmc_traffic.json (copy to profiles/ when it describes HEAD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o r02 -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/pmc_bench.json" 2> "$OLDPWD/gpurun_out/pmc.log" ); echo "exit $?" >> gpurun_out/pmc.log
rm -f gpurun_out/pmc_fetch_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch/r02_results.db gpurun_out/pmc_fetch_summary.txt '%k_pq_scan%' '%k_pq_prepass%' '%k_pq_rot%'
python tools/update_pmc_traffic.py gpurun_out/pmc_fetch/r02_results.db gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch
cat gpurun_out/pmc_fetch_summary.txt
