#!/bin/bash
# Round-5 GPU session 1: (1) bisect of the scan's FETCH_SIZE over four round-3/4 commits + HEAD (trees under _bisect/, built in the
# container), (2) TCC hit / miss / request counters of the scan at HEAD, (3) survivor pre-check of coarser (4-bit ...) tables with
# the measure build, (4) TCC / TCP counters of k_flat_gemm2.  Everything lands in gpurun_out/r05a_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
T=r05a

{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc; free -g | head -2; } > $O/${T}_env.txt 2>&1
( cd /tmp && timeout 120 rocprofv3 -L > $O/${T}_counters_avail.txt 2>&1 )
grep -o -E "\b(TCC|TCP|TCA|TA|TD)_[A-Z0-9_]+(\[[0-9]+\])?" $O/${T}_counters_avail.txt | sort -u > $O/${T}_counter_names.txt
wc -l $O/${T}_counter_names.txt

# un-profiled headline on this box
timeout 600 python bench.py --steps 20 --warmup 5 $FAST > $O/${T}_bench_fast.json 2> $O/${T}_bench_fast.log
python tools/show_bench.py $O/${T}_bench_fast.json

pmc_run() {   # $1 = tree dir, $2 = tag, $3 = counters, rest = command after `python`
  local dir=$1 tag=$2 ctr=$3; shift 3
  rm -rf $O/pmc_$tag
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$tag -o $tag -- python "$@" > /dev/null 2> $O/${T}_${tag}.log ); echo "exit $?" >> $O/${T}_${tag}.log
}

# (1) bisect
: > $O/${T}_bisect_fetch_size.md
for c in HEAD 95ac83a 27e639b c7f882f abaacf5 HEAD2; do
  if [ "$c" = HEAD ] || [ "$c" = HEAD2 ]; then dir=$R; else dir=$R/_bisect/$c; fi
  pmc_run $dir bis_$c FETCH_SIZE $dir/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs
  echo "## tree $c" >> $O/${T}_bisect_fetch_size.md
  python tools/pmc_summary.py $O/pmc_bis_$c/bis_${c}_results.db $O/${T}_bisect_fetch_size.md '%k_pq_scan_rot%'
  if [ "$c" = HEAD2 ]; then python tools/update_pmc_traffic.py $O/pmc_bis_$c/bis_${c}_results.db $O/pmc_traffic.json; fi
  rm -rf $O/pmc_bis_$c
done
cat $O/${T}_bisect_fetch_size.md | cut -c1-200

# (2) TCC counters of the scan at HEAD (aggregate and, when the db carries the instance dimension, per XCD)
pmc_run $R tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $R/bench.py --steps 2 --warmup 1 $FAST
python tools/pmc_summary.py $O/pmc_tcc/tcc_results.db $O/${T}_pmc_tcc_scan.md '%k_pq_scan_rot%'
python - <<EOF >> $O/${T}_pmc_tcc_scan.md 2>&1
import sqlite3
cur = sqlite3.connect("$O/pmc_tcc/tcc_results.db").cursor()
print("tables:", [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()][:60])
print("columns of counters_collection:", [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()])
for r in cur.execute("select * from counters_collection where kernel_name like '%k_pq_scan_rot%' limit 6").fetchall(): print(r)
EOF
rm -rf $O/pmc_tcc
pmc_run $R tcc2 "TCC_HIT[0] TCC_HIT[4] TCC_MISS[0] TCC_MISS[4]" $R/bench.py --steps 2 --warmup 1 $FAST
python tools/pmc_summary.py $O/pmc_tcc2/tcc2_results.db $O/${T}_pmc_tcc_scan.md '%k_pq_scan_rot%'
rm -rf $O/pmc_tcc2
cat $O/${T}_pmc_tcc_scan.md | cut -c1-300 | head -40

# (3) coarser tables: survivors per query on the bench index (measure build; results are wrong by design only in speed, not in ids)
: > $O/${T}_lut_step.txt
for st in 1 3 5 17; do
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LUT_STEP=$st timeout 600 python bench.py --steps 3 --warmup 1 $FAST > $O/${T}_lut_step$st.json 2> $O/${T}_lut_step$st.log
  python - <<EOF >> $O/${T}_lut_step.txt
import json
try:
    r = json.load(open("$O/${T}_lut_step$st.json"))
    print("lut_step", $st, "ms_per_step", r["ms_per_step"], "stage", r["stage_ms_per_step"], "survivors", r["filter_survivors_per_query"], "fallback_frac", r["certificate_fallback_fraction"])
except Exception as e:
    print("lut_step", $st, "failed", e)
EOF
done
cat $O/${T}_lut_step.txt

# (4) k_flat_gemm2 counters
pmc_run $R flat_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $R/tools/bench_configs.py flat --check 0 --steps 2
python tools/pmc_summary.py $O/pmc_flat_tcc/flat_tcc_results.db $O/${T}_pmc_flat_gemm2.md '%k_flat_gemm%'
rm -rf $O/pmc_flat_tcc
pmc_run $R flat_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" $R/tools/bench_configs.py flat --check 0 --steps 2
python tools/pmc_summary.py $O/pmc_flat_tcp/flat_tcp_results.db $O/${T}_pmc_flat_gemm2.md '%k_flat_gemm%'
rm -rf $O/pmc_flat_tcp
pmc_run $R flat_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" $R/tools/bench_configs.py flat --check 0 --steps 2
python tools/pmc_summary.py $O/pmc_flat_sq/flat_sq_results.db $O/${T}_pmc_flat_gemm2.md '%k_flat_gemm%'
rm -rf $O/pmc_flat_sq
cat $O/${T}_pmc_flat_gemm2.md | cut -c1-200
tail -n 3 $O/${T}_flat_tcp.log
ls -la $O | tail -n 5
