#!/bin/bash
# Flat 10M: sweep the threshold-phase size and the number of filtered stages (one box).  usage: flat_k1000_sweep.sh
run() {  # k, params...
  k=$1; shift; P=""; for a in "$@"; do P="$P --param $a"; done
  python tools/bench_configs.py flat --k $k --check 0 --steps 3 $P 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('k=$k $*', 'ms/step', r['ms_per_step'], 'scan', r['scan_ms'], 'select', r['select_ms'], 'finalize', r['finalize_ms'])"
}
run 1000 flat_pre_mult=32 flat_stages=5
run 1000 flat_pre_mult=32 flat_stages=6
run 1000 flat_pre_mult=16 flat_stages=5
run 1000 flat_pre_mult=32 flat_stages=4
run 10 flat_stages=1
run 10 flat_stages=2
run 10 flat_stages=3
run 100 flat_stages=2
run 100 flat_stages=4
run 100 flat_stages=5
