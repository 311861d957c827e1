#!/bin/bash
# Round-5 GPU session 21: ring depth of k_list_scan2's 16-query form (6 -> 4 / 3: three workgroups per CU) and of the 64-query form (3 -> 2)
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05u}
line() {  # $1 = label
python - <<PY >> $O/${T}_depth.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("$1:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("$1: failed", e)
PY
}
: > $O/${T}_depth.txt
for cfg in "4096 32" "8192 32"; do
for v in 0 4 3 0 4 3; do
  set -- $cfg $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS2_D=$3 timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist $1 nprobe $2 16-query form D=$3"
done
done
for cfg in "2048 64" "2048 32"; do
for v in 0 2 0 2; do
  set -- $cfg $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS2W_D=$3 timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist $1 nprobe $2 64-query form D=$3"
done
done
cat $O/${T}_depth.txt
tail -3 $O/${T}_tmp.log
