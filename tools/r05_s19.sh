#!/bin/bash
# Round-5 GPU session 19: k_list_scan3 ring depth / early issue (measure build), and the queries-per-group thresholds of the IVF-Flat scan
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05s}
line() {  # $1 = label
python - <<PY >> $O/${T}_scan3.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("$1:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("$1: failed", e)
PY
}
: > $O/${T}_scan3.txt
for v in "8 0" "4 0" "6 0" "9 0" "8 1" "9 1" "6 1" "8 0" "9 1"; do
  set -- $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$1 RSX_LS3_EARLY=$2 timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 128 scan3 D=$1 early=$2"
done
for v in "8 0" "9 1" "6 1"; do
  set -- $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$1 RSX_LS3_EARLY=$2 timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 64 --check 2 --steps 3 --param ivf_qtiles=8 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 64 scan3 D=$1 early=$2"
done
for cfg in "4096 32" "4096 64" "2048 32" "8192 128"; do
  for qt in 0 2 4; do
    set -- $cfg $qt
    timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 --param ivf_qtiles=$3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
    line "nlist $1 nprobe $2 ivf_qtiles=$3"
  done
done
cat $O/${T}_scan3.txt
tail -3 $O/${T}_tmp.log
