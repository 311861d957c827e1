cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() { tag=$1; shift; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs "$@" > gpurun_out/bench_p_$tag.json 2> gpurun_out/bench_p_$tag.log
  python -c "
import json; d=json.load(open('gpurun_out/bench_p_$tag.json')); s=d['stage_ms_per_step']; print('$tag', d['ms_per_step'], 'scan', s['scan'], 'scan0', s['scan0'], 'select', s['select'], 'fin', s['finalize'], 'fb', d['certificate_fallback_fraction'])"; }
run base
run chunk16k --param scan_chunk=16384
run chunk32k --param scan_chunk=32768
RSX_ROT_VARIANT=2 run chunk32k_noloop --param scan_chunk=32768
RSX_ROT_VARIANT=2 run base_noloop
RSX_ROT_VARIANT=6 run base_noloop_nostage
