#!/bin/bash
# Round-5 GPU session 3: bisect variant (abaacf5 with the byte-offset chunk bookkeeping restored), parameter A/Bs on HEAD, new GPU tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r05c
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
pmc_run() { local tag=$1 ctr=$2; shift 2; rm -rf /tmp/pmc_$tag
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$tag -o $tag -- python "$@" > /dev/null 2> $O/${T}_${tag}.log ); echo "exit $?" >> $O/${T}_${tag}.log; }
: > $O/${T}_bisect_fetch_size.md
for c in abaacf5_oldbk abaacf5 abaacf5_oldbk_again c7f882f; do
  dir=$R/_bisect/${c%_again}
  pmc_run bis_$c FETCH_SIZE $dir/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs
  echo "## tree $c" >> $O/${T}_bisect_fetch_size.md
  python tools/pmc_summary.py /tmp/pmc_bis_$c/bis_${c}_results.db $O/${T}_bisect_fetch_size.md '%k_pq_scan_rot%'
  rm -rf /tmp/pmc_bis_$c
done
grep -v "^# /" $O/${T}_bisect_fetch_size.md | cut -c1-160
# A/B on HEAD: table rows non-temporal (pq_pace bit 5), threshold sample rows, K'
: > $O/${T}_ab.txt
for p in "" "pq_pace=16544" "pq_pre_rows=2048" "pq_pre_rows=3072" "pq_fast_kp=64" "pq_pace=16544"  ""; do
  timeout 600 python bench.py --steps 20 --warmup 5 $FAST ${p:+--param $p} > $O/${T}_ab_tmp.json 2> $O/${T}_ab_tmp.log
  echo "== param '${p}'" >> $O/${T}_ab.txt
  python tools/show_bench.py $O/${T}_ab_tmp.json >> $O/${T}_ab.txt
  python -c "
import json,sys
r=json.loads([l for l in open('$O/${T}_ab_tmp.json') if l.startswith('{')][-1])
print('   survivors', r['filter_survivors_per_query'], 'fallback', r['certificate_fallback_fraction'])" >> $O/${T}_ab.txt
done
cat $O/${T}_ab.txt | cut -c1-330
# FETCH_SIZE with nt table rows
pmc_run nt FETCH_SIZE $R/bench.py --steps 2 --warmup 1 $FAST --param pq_pace=16544
echo "## HEAD + pq_pace bit 5 (nt table rows)" >> $O/${T}_bisect_fetch_size.md
python tools/pmc_summary.py /tmp/pmc_nt/nt_results.db $O/${T}_bisect_fetch_size.md '%k_pq_scan_rot%'
pmc_run head FETCH_SIZE $R/bench.py --steps 2 --warmup 1 $FAST
echo "## HEAD" >> $O/${T}_bisect_fetch_size.md
python tools/pmc_summary.py /tmp/pmc_head/head_results.db $O/${T}_bisect_fetch_size.md '%k_pq_scan_rot%'
tail -n 5 $O/${T}_bisect_fetch_size.md | cut -c1-160
# GPU suite (new tests included)
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
tail -n 12 $O/${T}_pytest_gpu.txt | cut -c1-250
