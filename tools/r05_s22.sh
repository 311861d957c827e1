#!/bin/bash
# Round-5 GPU session 22: evidence pass on the final sources: box facts, the whole GPU suite, smoke(), the FETCH_SIZE / MFMA-busy stamp, the full
# default bench line (stamp in place), rocprofv3 kernel stats of the headline and of IVF-Flat nlist 2048 / nprobe 128 (k = 10 and 1000), IVF-Flat nprobe 64
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export TAG=${TAG:-r05y}
bash tools/gpu_round.sh env tests_all smoke pmc_fetch
cp $O/pmc_traffic.json profiles/pmc_traffic.json
cat profiles/pmc_traffic.json | cut -c1-600
bash tools/gpu_round.sh bench prof ivfflat_prof
python tools/show_bench.py $O/${TAG}_bench_ivfpq100M.json
python - <<PY
import json
r=json.loads([l for l in open("$O/${TAG}_bench_ivfpq100M.json") if l.startswith("{")][-1])
print("roofline", {k:r["roofline"].get(k) for k in ("achieved","frac","traffic","traffic_over_algorithmic","mfma_busy","ms_per_launch")})
print("cpu", r["cpu_baseline"] and {k:r["cpu_baseline"][k] for k in ("value","cores","kind")}, "parity", r["cpu_parity_ids_and_scores_bit_exact"])
for k,v in (r.get("other_distributions") or {}).items(): print("dist", k, {x:v.get(x) for x in ("ms_per_step","exact_fallback_queries_per_step","filter_survivors_per_query","recall_at_10","oracle_parity_ids_and_scores","error")})
for k,v in (r.get("reference_n_docs_on_this_index") or {}).items(): print("ndocs", k, v["ms_per_step"], v["stage_ms"], v.get("oracle_parity_ids_and_scores"))
for k,v in (r.get("configs") or {}).items(): print("cfg", k, str(v)[:300])
print("one_call", r.get("one_call_all_queries")); print("recall", r.get("recall_at_10"), r.get("recall_informative"))
PY
cut -c1-1200 $O/${TAG}_ivfflat20M_nprobe128.json
head -n 30 $O/${TAG}_rocprof_stats_ivfflat20M_nprobe128.md | cut -c1-200
timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 64 --check 2 --steps 3 --ks 1000 > $O/${TAG}_ivfflat20M_nprobe64.json 2> $O/${TAG}_ivfflat64.log
cut -c1-1200 $O/${TAG}_ivfflat20M_nprobe64.json
