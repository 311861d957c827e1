TAG=r06s2 PYTEST_ARGS="tests/test_gpu_flat.py tests/test_gpu_scale.py" tools/gpu_round.sh tests
bash tools/flat_ab.sh old main > gpurun_out/r06s2_flat_ab.txt 2>&1
cut -c1-330 gpurun_out/r06s2_flat_ab.txt
for st in 3 4 6; do
  timeout 300 python tools/bench_configs.py flat --check 0 --steps 5 --k 1000 --param flat_stages=$st 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('stages $st', r.get('ms_per_step'), r.get('stage_ms'))
" | tee -a gpurun_out/r06s2_flat_stages.txt
done
timeout 900 python tools/bench_configs.py ivfflat --n 100000000 --ks 1000 --check 2 > gpurun_out/r06s2_ivfflat100M.json 2> gpurun_out/r06s2_ivfflat100M.log
python - <<'P'
import json
for l in open('gpurun_out/r06s2_ivfflat100M.json'):
    try: r = json.loads(l)
    except Exception: continue
    print(r.get('config'), r.get('ms_per_step'), r.get('stage_ms'))
    print('k1000', r['k1000'].get('ms_per_step'), r['k1000'].get('stage_ms'))
P
