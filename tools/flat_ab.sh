export TMPDIR=/tmp
for v in main s16 s24 s28 main; do
  lib=$PWD/retrieval-scaling_amd/csrc/librsx_$v.so; [ "$v" = main ] && lib=$PWD/retrieval-scaling_amd/csrc/librsx.so
  RSX_LIB=$lib timeout 600 python tools/bench_configs.py flat --check 4 --steps 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$v', r.get('config'), r.get('ms_per_step'), 'scan', r.get('scan_ms'), (r.get('roofline') or {}).get('frac'), r.get('oracle_parity_ids_and_scores'))
"
done
