#!/bin/bash
# Round-5 GPU session 17: IVF-Flat 20M, nlist 2048 / nprobe 128: 64-query groups (8 waves x 3 stages) against 32-query groups in the 8-wave form with 6 stages (measure build, RSX_LS2_FORM)
# (historical: the measure-build switches this session drove were removed with the experiment it measured; results: profiles/r05_ivfflat_wide.md)
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05q}
: > $O/${T}_form.txt
for cfg in "2048 128" "2048 64" "1024 128"; do
  for v in 0 1 0 1; do
    set -- $cfg $v
    RSX_LIB=$R/retrieval-scaling_amd/csrc/librsx_measure.so RSX_LS2_FORM=$3 timeout 600 python tools/bench_configs.py ivfflat --nlist $1 --nprobe $2 --check 2 --steps 3 > $O/${T}_tmp.json 2> $O/${T}_tmp.log
    python - <<PY >> $O/${T}_form.txt
import json
try:
    r=json.loads([l for l in open("$O/${T}_tmp.json") if l.startswith("{")][-1])
    print("nlist $1 nprobe $2 FORM=$3:", r["ms_per_step"], "scan", r["scan_ms"], "frac", r["roofline"]["frac"], "parity", r.get("oracle_parity_ids_and_scores"), "fb", r.get("certificate_fallback_queries_per_step"))
except Exception as e:
    print("nlist $1 nprobe $2 FORM=$3: failed", e)
PY
  done
done
cat $O/${T}_form.txt
