#!/bin/bash
# IVF-Flat at the reference's n_docs: rows per K' and closest lists in the threshold sample (one box)
run() {
  python tools/bench_configs.py ivfflat --k 1000 --check 0 --steps 3 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$*', '| ms/step', r['ms_per_step'], 'scan', r['scan_ms'], 'select', r['select_ms'], 'finalize', r['finalize_ms'], 'fallbacks', r['certificate_fallback_queries_per_step'])"
}
for cfg in "--nlist 2048 --nprobe 128" "--n 100000000"; do
  run $cfg --param ivf_pre_lists=2 --param ivf_pre_mult=4
  run $cfg --param ivf_pre_lists=2 --param ivf_pre_mult=2
  run $cfg --param ivf_pre_lists=2 --param ivf_pre_mult=1
  run $cfg --param ivf_pre_lists=4 --param ivf_pre_mult=1
done
