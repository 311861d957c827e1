#!/bin/bash
# Round-5 GPU session 15: the full default bench line (first run with the distribution legs inside bench.py), threshold-sample sweep at large k
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05o}
( time timeout 1500 python bench.py > $O/${T}_bench_full.json 2> $O/${T}_bench_full.log ) 2> $O/${T}_bench_full.time; echo "exit $?" >> $O/${T}_bench_full.log
tail -n 3 $O/${T}_bench_full.time; grep -c . $O/${T}_bench_full.json; tail -n 25 $O/${T}_bench_full.log | cut -c1-330
python tools/show_bench.py $O/${T}_bench_full.json
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_bench_full.json") if l.startswith("{")][-1])
print("roofline", {k:r["roofline"][k] for k in ("achieved","frac","traffic","traffic_over_algorithmic","mfma_busy","ms_per_launch")})
print("cpu", r["cpu_baseline"] and {k:r["cpu_baseline"][k] for k in ("value","cores","kind")}, "parity", r["cpu_parity_ids_and_scores_bit_exact"])
for k,v in (r.get("other_distributions") or {}).items(): print("dist", k, {x:v.get(x) for x in ("ms_per_step","exact_fallback_queries_per_step","filter_survivors_per_query","recall_at_10","oracle_parity_ids_and_scores","query_groups_per_probed_list","error")})
for k,v in (r.get("reference_n_docs_on_this_index") or {}).items(): print("ndocs", k, v["ms_per_step"], v["stage_ms"], v.get("oracle_parity_ids_and_scores"))
for k,v in (r.get("configs") or {}).items(): print("cfg", k, str(v)[:260])
print("one_call", r.get("one_call_all_queries")); print("recall", r.get("recall_at_10"), r.get("recall_informative"))
PY
: > $O/${T}_pre_sweep.txt
for pm in "160 16384" "80 16384" "80 8192" "40 8192" "160 8192"; do
  set -- $pm
  timeout 600 python tools/bench_configs.py largek --steps 5 --ks 100,1000 --check 0 --param pq_pre_mult=$1 --param pq_pre_max=$2 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  python - <<PY >> $O/${T}_pre_sweep.txt
import json
r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
for k,v in r.get("by_k", r).items():
    if isinstance(v, dict) and "ms_per_step" in v: print("pre_mult $1 pre_max $2", k, v["ms_per_step"], {x:v["stage_ms"][x] for x in ("scan0","scan","select","finalize")}, "fb", v.get("exact_fallback_queries_per_step"), "surv", v.get("filter_survivors_per_query"))
PY
done
cat $O/${T}_pre_sweep.txt | cut -c1-260
