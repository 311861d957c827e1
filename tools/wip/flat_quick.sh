#!/bin/bash
# Flat 10M, k = 10 / 100 / 1000 in one build (ms per batch, scan stage, parity spot check); extra args = --param name=value ...
python tools/bench_configs.py flat --k 10 --check 4 --steps 3 --ks 100,1000 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        print('$*', 'k=10', r['ms_per_step'], 'scan', r['scan_ms'], 'frac', r['roofline']['frac'], 'parity', r['oracle_parity_ids_and_scores'])
        for k in ('k100','k1000'): print('$*', k, r[k]['ms_per_step'], 'scan', r[k]['scan_ms'], 'select', r[k]['select_ms'], 'finalize', r[k]['finalize_ms'], 'parity', r[k]['oracle_parity_ids_and_scores'])"
