# round 3: pacing sweep of k_pq_scan_rot on the headline index + L2 counters at chosen settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
SETS="${SETS:---set pq_pace=0 --set pq_pace=1 --set pq_pace=2 --set pq_pace=3}"
timeout 900 python tools/exp_scan.py $SETS > gpurun_out/exp_pace.jsonl 2> gpurun_out/exp_pace.log; echo "exit $?" >> gpurun_out/exp_pace.log
OUT=gpurun_out/r03_pace_pmc.txt; rm -f $OUT
for P in ${PMC_PACES:-1 2}; do
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OLDPWD/gpurun_out/pmc_pace" -o c -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs --param pq_pace=$P > "$OLDPWD/gpurun_out/pacepmc.json" 2> "$OLDPWD/gpurun_out/pacepmc.log" ); echo "exit $?" >> gpurun_out/pacepmc.log
echo "## pq_pace=$P" >> $OUT
python tools/pmc_summary.py gpurun_out/pmc_pace/c_results.db $OUT '%k_pq_scan_rot%'; rm -rf gpurun_out/pmc_pace
done
cat gpurun_out/exp_pace.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['set'], r['round'], r['qps'], r['ms_per_step'], r['stages']['scan'], r['fallback_queries'], r['same_as_first'])
"
cat $OUT; tail -n 2 gpurun_out/exp_pace.log
