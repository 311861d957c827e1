TAG=r06s8 PYTEST_ARGS="tests/test_gpu_ivf.py tests/test_gpu_scale.py tests/test_gpu_random.py" PYTEST_K="large_k or n_docs or ties or random_ivfpq or final" tools/gpu_round.sh tests_all
for v in oldsel main oldsel main; do
  lib=$PWD/retrieval-scaling_amd/csrc/librsx_$v.so; [ "$v" = main ] && lib=$PWD/retrieval-scaling_amd/csrc/librsx.so
  RSX_LIB=$lib timeout 600 python tools/bench_configs.py largek --ks 100,1000,2000 --steps 5 --check 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    for kk, x in r['by_k'].items():
        print('$v', kk, x['ms_per_step'], x['stage_ms'].get('finalize'), x['stage_ms'].get('select'), x['stage_ms'].get('scan0'), x.get('oracle_parity_ids_and_scores'), x.get('exact_fallback_queries_per_step'))
" | tee -a gpurun_out/r06s8_largek_ab.txt
done
