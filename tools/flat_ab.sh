# Flat 10M A/B of engine builds (tools/build_variant.sh): tools/flat_ab.sh [variant ...]   (main = librsx.so); k = 10 and 1000
export TMPDIR=/tmp
for v in ${@:-old main old main}; do
  lib=$PWD/retrieval-scaling_amd/csrc/librsx_$v.so; [ "$v" = main ] && lib=$PWD/retrieval-scaling_amd/csrc/librsx.so
  RSX_LIB=$lib timeout 600 python tools/bench_configs.py flat --check 4 --steps 5 --ks 1000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    for key, x in (('k10', r), ('k1000', r.get('k1000') or {})):
        print('$v', key, r.get('config'), x.get('ms_per_step'), 'scan', x.get('scan_ms'), 'select', x.get('select_ms'), 'fin', x.get('finalize_ms'), (x.get('roofline') or {}).get('frac'), x.get('oracle_parity_ids_and_scores'), x.get('stage_ms'))
"
done
