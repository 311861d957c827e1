#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python tools/exp_scan.py --rounds 2 --steps 10 --set "" --set pq_pre_rows=2048 --set pq_pre_rows=3072 > gpurun_out/${TAG:-r03s}_sweep.jsonl 2> gpurun_out/${TAG:-r03s}_sweep.log
python - gpurun_out/${TAG:-r03s}_sweep.jsonl <<'P'
import json,sys
for l in open(sys.argv[1]):
    r=json.loads(l); print(repr(r['set']), r['round'], r['ms_per_step'], {k:v for k,v in r['stages'].items() if k in('scan0','scan','select','finalize','total')}, 'cand', r['cand_mean'], 'fb', r['fallback_queries'], r['same_as_first'])
P
