#!/bin/bash
# Round-5 GPU session 16: prefetch wave in k_list_scan2 (measure build switches): IVF-Flat 20M at nlist 2048 / nprobe 128 (64-query groups) and nlist 4096 / nprobe 32 (16-query groups)
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05p}
: > $O/${T}_pf.txt
for cfg in "2048 128" "4096 32"; do
  set -- $cfg
  for v in "0 0" "2 4" "4 4" "2 8" "4 8" "2 2" "0 0"; do
    set -- $cfg $v
    RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS2_PF=$3 RSX_LS2_PD=$4 timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
    python - <<PY >> $O/${T}_pf.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("nlist $1 nprobe $2 PF=$3 PD=$4:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("nlist $1 nprobe $2 PF=$3 PD=$4: failed", e)
PY
  done
done
cat $O/${T}_pf.txt
