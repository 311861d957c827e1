EXP_ARGS="--layouts 2 --rounds 2 --steps 10" tools/ab_variants.sh main v32 v64 v34 v66 main 2>&1 | tee gpurun_out/r06s4_scan_lockstep.txt
for pp in "ivf_pre_lists=0" "ivf_pre_lists=1" "ivf_pre_mult=2" "ivf_pre_mult=1" "ivf_pre_lists=1 --param ivf_pre_mult=2"; do
  timeout 300 python tools/bench_configs.py ivfflat --n 20000000 --nlist 2048 --nprobe 128 --check 0 --steps 5 --k 1000 --param $pp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$pp', r.get('ms_per_step'), r.get('filter_keys_per_query'), r.get('stage_ms'))
" | tee -a gpurun_out/r06s4_ivfflat_pre.txt
done
