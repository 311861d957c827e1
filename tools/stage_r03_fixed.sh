#!/bin/bash
# fixed-cost study on one GPU box: per-kernel rocprof stats of the bench loop, sample-size sweep on the headline index, K' sweep at the
# reference's M = 16 operating point -> gpurun_out/${TAG}_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03f}
BARGS="--steps 5 --warmup 2 --cpu-queries 0 --no-recall --no-configs --no-faiss"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o $TAG -- python "$OLDPWD/bench.py" $BARGS > "$OLDPWD/gpurun_out/${TAG}_prof_bench.json" 2> "$OLDPWD/gpurun_out/${TAG}_prof.log" ); echo "exit $?" >> gpurun_out/${TAG}_prof.log
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py $BARGS)"
rm -rf gpurun_out/prof
timeout 600 python tools/exp_scan.py --rounds 1 --steps 8 --set "" --set pq_pre_rows=4096 --set pq_pre_rows=8192 --set pq_pre_rows=16384 --set pq_fast_kp=64 > gpurun_out/${TAG}_sweep_headline.jsonl 2> gpurun_out/${TAG}_sweep_headline.log; echo "exit $?" >> gpurun_out/${TAG}_sweep_headline.log
timeout 600 python tools/exp_scan.py --m 16 --nlist 8192 --nprobe 512 --rounds 1 --steps 3 --warmup 2 --set "" --set pq_fast_kp=256 --set pq_fast_kp=512 --set pq_fast_kp=1024 --set pq_pre_rows=8192 --set pq_pre_rows=8192,pq_fast_kp=512 > gpurun_out/${TAG}_sweep_m16.jsonl 2> gpurun_out/${TAG}_sweep_m16.log; echo "exit $?" >> gpurun_out/${TAG}_sweep_m16.log
cat gpurun_out/${TAG}_sweep_headline.jsonl gpurun_out/${TAG}_sweep_m16.jsonl | cut -c1-700
sed -n 1,40p gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md | cut -c1-120
