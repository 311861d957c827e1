#!/bin/bash
# Round-3 evidence pass (same recipe as round 2) on one GPU box: rocprofv3 stats, PMC FETCH_SIZE (stamped traffic file), SQ counters of the scan kernel,
# the full default bench line, and the config-5 per-rank workload (125M vectors on one GPU).  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r03}
BARGS="--steps 5 --warmup 2 --cpu-queries 0 --no-recall --no-configs"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o $TAG -- python "$OLDPWD/bench.py" $BARGS > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.log" ); echo "exit $?" >> gpurun_out/prof.log
python tools/rocprof_summary.py gpurun_out/prof/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py $BARGS)"
rm -rf gpurun_out/prof
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/pmc_bench.json" 2> "$OLDPWD/gpurun_out/pmc.log" ); echo "exit $?" >> gpurun_out/pmc.log
rm -f gpurun_out/${TAG}_pmc_fetch_size.md
python tools/pmc_summary.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/${TAG}_pmc_fetch_size.md '%k_pq_scan%' '%k_pq_prepass%' '%k_pq_rot%'
python tools/update_pmc_traffic.py gpurun_out/pmc_fetch/${TAG}_results.db gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmc_fetch
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d "$OLDPWD/gpurun_out/pmc_sq" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/pmc_sq.json" 2> "$OLDPWD/gpurun_out/pmc_sq.log" ); echo "exit $?" >> gpurun_out/pmc_sq.log
rm -f gpurun_out/${TAG}_pmc_sq_counters.md
python tools/pmc_summary.py gpurun_out/pmc_sq/${TAG}_results.db gpurun_out/${TAG}_pmc_sq_counters.md '%k_pq_scan%' '%k_pq_rot%' '%k_finalize%' '%k_pq_lut%'
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$OLDPWD/gpurun_out/pmc_sq2" -o $TAG -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > /dev/null 2> "$OLDPWD/gpurun_out/pmc_sq2.log" ); echo "exit $?" >> gpurun_out/pmc_sq2.log
python tools/pmc_summary.py gpurun_out/pmc_sq2/${TAG}_results.db gpurun_out/${TAG}_pmc_sq_counters.md '%k_pq_scan%'
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_ivfpq100M.json 2> gpurun_out/${TAG}_bench.log; echo "exit $?" >> gpurun_out/${TAG}_bench.log
: > gpurun_out/${TAG}_per_rank_workloads.txt
for cfg in "50000000 1024" "25000000 1024" "12500000 1024"; do
  set -- $cfg
  timeout 600 python bench.py --n $1 --batch $2 --steps 20 --warmup 5 --cpu-queries 0 --no-recall --no-configs > gpurun_out/${TAG}_n$1_b$2.json 2> gpurun_out/${TAG}_n$1_b$2.log
  python - gpurun_out/${TAG}_n$1_b$2.json $1 $2 >> gpurun_out/${TAG}_per_rank_workloads.txt <<'P'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=j["stage_ms_per_step"]
print(f"n={sys.argv[2]} batch={sys.argv[3]}: {j['ms_per_step']:.3f} ms/step, scan {s['scan']:.3f}, fixed {j['ms_per_step']-s['scan']:.3f}, {j['value']:.0f} q/s of this one rank")
P
done
timeout 600 python bench.py --n 125000000 --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs > gpurun_out/${TAG}_bench_125M_one_rank_of_config5.json 2> gpurun_out/${TAG}_bench_125M.log; echo "exit $?" >> gpurun_out/${TAG}_bench_125M.log
timeout 500 python tools/bench_configs.py latency > gpurun_out/${TAG}_latency_ivfpq100M.json 2> gpurun_out/${TAG}_latency.log; echo "exit $?" >> gpurun_out/${TAG}_latency.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_flat" -o $TAG -- python "$OLDPWD/tools/bench_configs.py" flat --check 64 --steps 3 > "$OLDPWD/gpurun_out/${TAG}_flat10M.json" 2> "$OLDPWD/gpurun_out/${TAG}_flat10M.log" ); echo "exit $?" >> gpurun_out/${TAG}_flat10M.log
python tools/rocprof_summary.py gpurun_out/prof_flat/${TAG}_results.db gpurun_out/${TAG}_rocprof_stats_flat10M.md "Flat 10M x 768 batch 1024 (tools/bench_configs.py flat --check 64 --steps 3)"
rm -rf gpurun_out/prof_flat
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$OLDPWD/gpurun_out/pmc_flat" -o $TAG -- python "$OLDPWD/tools/bench_configs.py" flat --check 0 --steps 2 > /dev/null 2> "$OLDPWD/gpurun_out/${TAG}_pmc_flat.log" ); echo "exit $?" >> gpurun_out/${TAG}_pmc_flat.log
rm -f gpurun_out/${TAG}_pmc_sq_flat_gemm2.md
python tools/pmc_summary.py gpurun_out/pmc_flat/${TAG}_results.db gpurun_out/${TAG}_pmc_sq_flat_gemm2.md '%k_flat_gemm2%'
rm -rf gpurun_out/pmc_flat
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.txt
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; python -c "import faiss" 2>&1 | tail -1; } > gpurun_out/${TAG}_gpu_box_env.txt 2>&1
ls -la gpurun_out | tail -20
for f in gpurun_out/*.log; do echo "== $f"; tail -n 3 "$f"; done
