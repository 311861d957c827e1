#!/bin/bash
# Round-5 GPU session 20: IVF tests on the new group-size selection; k_list_scan3 K-step stages (D = 3 / 4) against whole-row stages (ROWS, D = 3 / 4 / 5)
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05t}
timeout 1200 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_scale.py -q -m gpu -x 2>&1 | tail -5 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
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
for cfg in "2048 128" "1024 128"; do
for v in "0 0" "3 0" "4 1" "3 1" "5 1" "0 0" "4 1"; do
  set -- $cfg $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$3 RSX_LS3_ROWS=$4 timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist $1 nprobe $2 scan3 D=$3 rows=$4"
done
done
for v in "0 0" "4 1"; do
  set -- $v
  RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS3_D=$1 RSX_LS3_ROWS=$2 timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 64 --check 2 --steps 3 --param ivf_qtiles=8 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
  line "nlist 2048 nprobe 64 ivf_qtiles=8 scan3 D=$1 rows=$2"
done
timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 64 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
line "nlist 2048 nprobe 64 default"
cat $O/${T}_scan3.txt
tail -3 $O/${T}_tmp.log
