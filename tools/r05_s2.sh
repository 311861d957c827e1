#!/bin/bash
# Round-5 GPU session 2: finer FETCH_SIZE bisect (c7f882f .. abaacf5), per-XCD TCC hit / miss of the scan, GPU suite on the pruned build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=r05b
pmc_run() { local tag=$1 ctr=$2; shift 2; rm -rf $O/pmc_$tag
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$tag -o $tag -- python "$@" > /dev/null 2> $O/${T}_${tag}.log ); echo "exit $?" >> $O/${T}_${tag}.log; }
: > $O/${T}_bisect_fetch_size.md
for c in c7f882f 414d17b abcd82b abaacf5 c7f882f_again abaacf5_again; do
  dir=$R/_bisect/${c%_again}
  pmc_run bis_$c FETCH_SIZE $dir/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs
  echo "## tree $c" >> $O/${T}_bisect_fetch_size.md
  python tools/pmc_summary.py $O/pmc_bis_$c/bis_${c}_results.db $O/${T}_bisect_fetch_size.md '%k_pq_scan_rot%'
  rm -rf $O/pmc_bis_$c
done
cat $O/${T}_bisect_fetch_size.md | grep -v "^#  */" | cut -c1-160
# per-XCD TCC counters (raw counters carry DIMENSION_INSTANCE[0:15] x DIMENSION_XCC[0:7])
pmc_run tccx "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" $R/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs --no-faiss
python - <<PY > $O/${T}_pmc_tcc_per_xcd.md 2>&1
import sqlite3, collections
db = "$O/pmc_tccx/tccx_results.db"
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
print("# tables/views:", [t.split('_0000')[0] for t in tabs])
for t in tabs:
    if 'pmc' in t.lower() or 'dim' in t.lower():
        cols = [r[1] for r in cur.execute(f"pragma table_info('{t}')").fetchall()]
        print("##", t.split('_0000')[0], cols)
        for r in cur.execute(f"select * from '{t}' limit 3").fetchall(): print("   ", str(r)[:300])
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
print("# counters_collection columns:", cols)
rows = cur.execute("select * from counters_collection where kernel_name like '%k_pq_scan_rot%' limit 2").fetchall()
for r in rows: print(str(r)[:1500])
PY
head -c 6000 $O/${T}_pmc_tcc_per_xcd.md
cp $O/pmc_tccx/tccx_results.db $O/${T}_tccx.db 2>/dev/null; ls -la $O/${T}_tccx.db
rm -rf $O/pmc_tccx
# GPU suite on the pruned build
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
tail -n 15 $O/${T}_pytest_gpu.txt | cut -c1-250
