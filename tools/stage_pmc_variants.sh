cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for chunk in 8192 16384 32768; do
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o v -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs --param scan_chunk=$chunk > /dev/null 2> "$OLDPWD/gpurun_out/pmc.log" )
rm -f gpurun_out/pmc_v.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch/v_results.db gpurun_out/pmc_v.txt '%k_pq_scan_rot%'
echo "chunk $chunk: $(grep FETCH gpurun_out/pmc_v.txt)"
rm -rf gpurun_out/pmc_fetch
done
