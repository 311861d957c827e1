#!/bin/bash
# Round-5 GPU session 25: k_list_scan3 with the bias in the stages (squared distances) and at d = 384 / 1024: parity tests, then L2 A/B at nlist 2048 / nprobe 128
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05x}
timeout 900 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_scale.py -q -m gpu -x 2>&1 | tail -5 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
line() {
python - <<PY >> $O/${T}_l2.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("$1:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("$1: failed", e)
PY
}
: > $O/${T}_l2.txt
for v in "l2 8" "l2 4" "ip 8" "l2 8" "l2 4"; do
  set -- $v
  timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 --metric $1 --param ivf_qtiles=$2 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 128 metric $1 ivf_qtiles=$2"
done
cat $O/${T}_l2.txt
tail -3 $O/${T}_tmp.log
