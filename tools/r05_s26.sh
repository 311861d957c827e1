#!/bin/bash
# Round-5 GPU session 26 (last): re-stamp FETCH_SIZE / MFMA busy on the final sources, the full default bench line with the stamp in it, the whole GPU suite
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export TAG=${TAG:-r05yy}
bash tools/gpu_round.sh pmc_fetch
cp $O/pmc_traffic.json profiles/pmc_traffic.json
grep -E "hbm_bytes|source_sha|mfma_busy_frac" profiles/pmc_traffic.json
bash tools/gpu_round.sh bench
python tools/show_bench.py $O/${TAG}_bench_ivfpq100M.json
python - <<PY
import json
r=json.loads([l for l in open("$O/${TAG}_bench_ivfpq100M.json") if l.startswith("{")][-1])
print("roofline", {k:r["roofline"].get(k) for k in ("achieved","frac","traffic","traffic_over_algorithmic","mfma_busy","ms_per_launch")})
print("cpu", r["cpu_baseline"] and {k:r["cpu_baseline"][k] for k in ("value","cores","kind")}, "parity", r["cpu_parity_ids_and_scores_bit_exact"])
for k,v in (r.get("configs") or {}).items(): print("cfg", k, v.get("ms_per_step"), v.get("scan_ms"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), {kk:vv.get("ms_per_step") for kk,vv in v.items() if isinstance(vv, dict) and "ms_per_step" in vv})
PY
PYTEST_ARGS="" bash tools/gpu_round.sh tests_all smoke
cat $O/${TAG}_smoke.log | tail -2
