# round 3: same-box A/B of librsx_head.so (previous build, copied by hand) vs the current build, then PMC passes of the current one
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_random.py -q -m gpu -x --timeout 300 -p no:cacheprovider > gpurun_out/ab_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/ab_pytest.log; tail -3 gpurun_out/ab_pytest.log
B="--steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs"
for r in 1 2; do
  RSX_LIB=$PWD/retrieval-scaling_amd/csrc/librsx_head.so timeout 600 python bench.py $B > gpurun_out/ab_head$r.json 2> gpurun_out/ab_head$r.log
  timeout 600 python bench.py $B > gpurun_out/ab_new$r.json 2> gpurun_out/ab_new$r.log
done
OUT=gpurun_out/r03_ab_pmc.txt; rm -f $OUT
run() { name=$1; shift
  ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc "$@" -d "$OLDPWD/gpurun_out/pmc_$name" -o c -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/abpmc_$name.json" 2> "$OLDPWD/gpurun_out/abpmc_$name.log" ); echo "exit $?" >> gpurun_out/abpmc_$name.log
  echo "## $name: $*" >> $OUT
  python tools/pmc_summary.py gpurun_out/pmc_$name/c_results.db $OUT '%k_pq_scan_rot%'
  rm -rf gpurun_out/pmc_$name
}
run rd TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
python - <<'PY' >> $OUT
import json
for n in ("ab_head1","ab_new1","ab_head2","ab_new2"):
    try:
        r=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
        print(n, r["value"], r["ms_per_step"], r["stage_ms_per_step"]["scan"], r["certificate_fallback_fraction"])
    except Exception as e: print(n, "failed", e)
PY
cat $OUT
