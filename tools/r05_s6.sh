#!/bin/bash
# Round-5 GPU session 6: GPU suite on the finalize / final_tab / gather changes, phase traces, headline + large k
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05f}
FAST="--cpu-queries 0 --no-recall --no-configs --no-faiss"
timeout 2400 python -m pytest tests -q -m gpu -x --timeout 900 -p no:cacheprovider > $O/${T}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${T}_pytest_gpu.txt
tail -n 12 $O/${T}_pytest_gpu.txt | cut -c1-250
RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so timeout 600 python tools/exp_ft_trace.py 100000000 10 100 1000 > $O/${T}_ft_trace.txt 2> $O/${T}_ft_trace.log; echo "exit $?" >> $O/${T}_ft_trace.log
cat $O/${T}_ft_trace.txt | cut -c1-400; tail -n 2 $O/${T}_ft_trace.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 $FAST > $O/${T}_bench_fast.json 2> $O/${T}_bench_fast.log
python tools/show_bench.py $O/${T}_bench_fast.json
timeout 900 python tools/bench_configs.py largek --steps 5 > $O/${T}_largek.json 2> $O/${T}_largek.log; echo "exit $?" >> $O/${T}_largek.log
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_largek.json") if l.startswith("{")][-1])
for k,v in r.get("by_k", r).items():
    if isinstance(v, dict) and "ms_per_step" in v: print(k, v["ms_per_step"], v.get("stage_ms"), "fb", v.get("exact_fallback_queries_per_step"), "parity", v.get("oracle_parity_ids_and_scores"))
PY
timeout 900 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 --ks 1000 > $O/${T}_ivfflat20M.json 2> $O/${T}_ivfflat20M.log
python - <<PY
import json
r=json.loads([l for l in open("$O/${T}_ivfflat20M.json") if l.startswith("{")][-1]); k=r.get("k1000",{})
print("ivfflat20M", {x:r[x] for x in ("ms_per_step","scan_ms","select_ms","finalize_ms")}, "k1000", {x:k.get(x) for x in ("ms_per_step","scan_ms","select_ms","finalize_ms","certificate_fallback_queries_per_step","oracle_parity_ids_and_scores")})
PY
