#!/bin/bash
# Round-5 GPU session 24: k_list_scan3 with KS K steps per stage (= per barrier) and D stages: (4,1) default, (3,2) (4,2) (3,3) (2,4) (2,2) (2,3)
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05w}
line() {
python - <<PY >> $O/${T}_ks.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("$1:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("$1: failed", e)
PY
}
: > $O/${T}_ks.txt
for v in "0 1" "3 2" "4 2" "3 3" "2 4" "2 2" "2 3" "0 1" "3 2"; do
  set -- $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$1 RSX_LS3_KS=$2 timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 128 scan3 D=$1 KS=$2"
done
for v in "0 1" "3 2" "2 4" "3 3"; do
  set -- $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$1 RSX_LS3_KS=$2 timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 64 --check 2 --steps 3 --param ivf_qtiles=8 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 64 ivf_qtiles=8 scan3 D=$1 KS=$2"
done
cat $O/${T}_ks.txt
tail -3 $O/${T}_tmp.log
