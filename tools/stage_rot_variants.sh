cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for v in ${VARIANTS:-0 1 2 6}; do
  RSX_ROT_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs $BENCH_ARGS > gpurun_out/bench_rv$v.json 2> gpurun_out/bench_rv$v.log
  python -c "
import json; d=json.load(open('gpurun_out/bench_rv$v.json')); print('variant $v', d['ms_per_step'], d['stage_ms_per_step']['scan'])"
done
