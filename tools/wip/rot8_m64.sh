#!/bin/bash
# M = 64 on 100M vectors (nlist 4096, nprobe 32, batch 1024): 4 against 8 queries per pass (pq_rot8), one box
for p in 0 1 0 1; do
  python tools/bench_configs.py largek --m 64 --ks 10,1000 --steps 5 --check 2 --param pq_rot8=$p 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('pq_rot8=$p', r['config'], {k:(v['ms_per_step'], 'scan', v['stage_ms']['scan'], 'fb', v['exact_fallback_queries_per_step'], v.get('oracle_parity_ids_and_scores')) for k,v in r['by_k'].items()})"
done
