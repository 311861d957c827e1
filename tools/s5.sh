for pp in "flat_pre_mult=32" "flat_pre_mult=8 --param flat_pre_unit=16384" "flat_pre_mult=16 --param flat_pre_unit=16384" "flat_pre_mult=8 --param flat_pre_unit=16384 --param flat_stages=5" "flat_pre_mult=4 --param flat_pre_unit=8192" "flat_pre_mult=16 --param flat_pre_unit=32768"; do
  timeout 300 python tools/bench_configs.py flat --check 4 --steps 5 --k 1000 --param $pp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$pp', r.get('ms_per_step'), r.get('oracle_parity_ids_and_scores'), r.get('certificate_fallback_queries_per_step'), r.get('stage_ms'))
" | tee -a gpurun_out/r06s5_flat_k1000_pre.txt
done
for pp in "flat_pre_unit=0" "flat_pre_unit=8192" "flat_pre_unit=4096" "flat_pre_unit=65536"; do
  timeout 300 python tools/bench_configs.py flat --check 4 --steps 5 --k 10 --param $pp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('k10 $pp', r.get('ms_per_step'), r.get('oracle_parity_ids_and_scores'), r.get('certificate_fallback_queries_per_step'), r.get('stage_ms'))
" | tee -a gpurun_out/r06s5_flat_k1000_pre.txt
done
