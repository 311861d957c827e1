# round 3, calibration of the PMC traffic figure of k_pq_scan_rot: FETCH_SIZE (and L2 hit/miss, EA read requests) at batch 128
# (about one 4-query group per probed list: every list is read ONCE, so bytes == group bytes == ~unique bytes: the counter's
# scale factor for THIS kernel's b128 + b64 access mix falls out) and at batch 1024 (2 groups per list).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/r03_calib.txt; rm -f $OUT
run() { # name batch counters...
  name=$1; batch=$2; shift 2
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc "$@" -d "$OLDPWD/gpurun_out/pmc_$name" -o c -- python "$OLDPWD/bench.py" --batch $batch --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/calib_$name.json" 2> "$OLDPWD/gpurun_out/calib_$name.log" ); echo "exit $?" >> gpurun_out/calib_$name.log
  echo "## $name: batch $batch, counters $*" >> $OUT
  python tools/pmc_summary.py gpurun_out/pmc_$name/c_results.db $OUT '%k_pq_scan_rot%'
  python - "$name" >> $OUT <<'PY'
import json, sys
try:
    r = json.loads(open(f"gpurun_out/calib_{sys.argv[1]}.json").read().strip().splitlines()[-1])["roofline"]
    print(f"unique_bytes={r['algorithmic_bytes_per_launch']:.6g} group_bytes={r['lds']['achieved']*1e9*r['ms_per_launch']*1e-3/4:.6g} ms_per_launch={r['ms_per_launch']}")
except Exception as e:
    print("no bench line:", e)
PY
  rm -rf gpurun_out/pmc_$name
}
run f128 128 FETCH_SIZE
run f1024 1024 FETCH_SIZE
run h1024 1024 TCC_HIT_sum TCC_MISS_sum
run r1024 1024 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run r128 128 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cat $OUT
