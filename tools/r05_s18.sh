#!/bin/bash
# Round-5 GPU session 18: k_list_scan3 (query-stationary IVF-Flat scan, 128 queries per group): parity tests, then A/B against the 64-query form
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05r}
timeout 900 python -m pytest tests/test_gpu_ivf.py -q -m gpu -x -k "query_stationary or engine_knobs or fp16" 2>&1 | tail -5 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
: > $O/${T}_scan3.txt
for cfg in "2048 128" "1024 128" "2048 64"; do
  for v in 8 4 8 4 2; do
    set -- $cfg $v
    timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 --param ivf_qtiles=$3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
    python - <<PY >> $O/${T}_scan3.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("nlist $1 nprobe $2 ivf_qtiles=$3:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("nlist $1 nprobe $2 ivf_qtiles=$3: failed", e)
PY
  done
done
cat $O/${T}_scan3.txt
tail -3 $O/${T}_tmp.log
