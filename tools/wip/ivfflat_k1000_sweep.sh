#!/bin/bash
# IVF-Flat at the reference's n_docs: closest lists in the threshold sample (0 = per query, as many as needed; one box)
run() {
  python tools/bench_configs.py ivfflat --k 1000 --check 2 --steps 3 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$*', '| ms/step', r['ms_per_step'], 'scan', r['scan_ms'], 'select', r['select_ms'], 'finalize', r['finalize_ms'], 'fallbacks', r['certificate_fallback_queries_per_step'], 'parity', r.get('oracle_parity_ids_and_scores'))"
}
for cfg in "--nlist 2048 --nprobe 128" "--n 100000000"; do
  for pl in 2 0; do run $cfg --param ivf_pre_lists=$pl; done
done
