#!/bin/bash
# One GPU-box session: environment facts, GPU tests, smoke, bench, rocprof. Everything is logged under
# gpurun_out/ so a single call returns as much evidence as possible.  Usage: tools/gpu_round.sh [stage ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${*:-env tests smoke bench_small}"
for s in $STAGES; do
  case $s in
    env)
      { rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; python -c "import faiss" 2>&1 | tail -1; } > gpurun_out/env.log 2>&1 ;;
    tests)
      timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log ;;
    tests_all)
      timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log ;;
    bench_small)
      timeout 900 python bench.py --n 4000000 --nlist 1024 --steps 5 --warmup 2 --cpu-queries 32 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.log; echo "exit $?" >> gpurun_out/bench_small.log ;;
    bench)
      timeout 1700 python bench.py --steps 10 --warmup 3 --ab > gpurun_out/bench.json 2> gpurun_out/bench.log; echo "exit $?" >> gpurun_out/bench.log ;;
    diag)
      timeout 300 python tools/diag_synth.py > gpurun_out/diag_synth.log 2>&1 ;;
    configs)
      timeout 900 python tools/bench_configs.py flat > gpurun_out/cfg_flat.json 2> gpurun_out/cfg_flat.log; echo "exit $?" >> gpurun_out/cfg_flat.log
      timeout 900 python tools/bench_configs.py ivfflat > gpurun_out/cfg_ivfflat.json 2> gpurun_out/cfg_ivfflat.log; echo "exit $?" >> gpurun_out/cfg_ivfflat.log ;;
    ivfflat100m)
      timeout 1500 python tools/bench_configs.py ivfflat --n 100000000 > gpurun_out/cfg_ivfflat100m.json 2> gpurun_out/cfg_ivfflat100m.log; echo "exit $?" >> gpurun_out/cfg_ivfflat100m.log ;;
    latency_ab)
      timeout 900 python tools/bench_configs.py latency --param pq_filter=0 > gpurun_out/cfg_latency_ab.json 2> gpurun_out/cfg_latency_ab.log; echo "exit $?" >> gpurun_out/cfg_latency_ab.log ;;
    latency)
      timeout 900 python tools/bench_configs.py latency > gpurun_out/cfg_latency.json 2> gpurun_out/cfg_latency.log; echo "exit $?" >> gpurun_out/cfg_latency.log ;;
    pmc_sq)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d "$OLDPWD/gpurun_out/pmc_sq_flat" -o r01 -- python "$OLDPWD/tools/bench_configs.py" flat --n 2000000 --check 0 --steps 2 > "$OLDPWD/gpurun_out/pmc_sq_flat.json" 2> "$OLDPWD/gpurun_out/pmc_sq_flat.log" ); echo "exit $?" >> gpurun_out/pmc_sq_flat.log
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d "$OLDPWD/gpurun_out/pmc_sq_pq" -o r01 -- python "$OLDPWD/bench.py" --n 20000000 --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/pmc_sq_pq.json" 2> "$OLDPWD/gpurun_out/pmc_sq_pq.log" ); echo "exit $?" >> gpurun_out/pmc_sq_pq.log
      rm -f gpurun_out/pmc_sq_summary.txt
      python tools/pmc_summary.py gpurun_out/pmc_sq_flat/r01_results.db gpurun_out/pmc_sq_summary.txt '%k_flat_gemm%' '%k_select%'
      python tools/pmc_summary.py gpurun_out/pmc_sq_pq/r01_results.db gpurun_out/pmc_sq_summary.txt '%k_pq_scan8%' '%k_pq_lut%' '%k_finalize%'
      rm -rf gpurun_out/pmc_sq_flat gpurun_out/pmc_sq_pq ;;
    cfg_flat_only)
      timeout 900 python tools/bench_configs.py flat > gpurun_out/cfg_flat.json 2> gpurun_out/cfg_flat.log; echo "exit $?" >> gpurun_out/cfg_flat.log ;;
    prof_flat)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_flat" -o r01 -- python "$OLDPWD/tools/bench_configs.py" flat --check 0 --steps 3 > "$OLDPWD/gpurun_out/prof_flat.json" 2> "$OLDPWD/gpurun_out/prof_flat.log" ); echo "exit $?" >> gpurun_out/prof_flat.log ;;
    prof_ivfflat)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_ivfflat" -o r01 -- python "$OLDPWD/tools/bench_configs.py" ivfflat --check 0 --steps 3 > "$OLDPWD/gpurun_out/prof_ivfflat.json" 2> "$OLDPWD/gpurun_out/prof_ivfflat.log" ); echo "exit $?" >> gpurun_out/prof_ivfflat.log ;;
    pmc_ivfflat)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_ivfflat" -o r01 -- python "$OLDPWD/tools/bench_configs.py" ivfflat --check 0 --steps 2 > "$OLDPWD/gpurun_out/pmc_ivfflat.json" 2> "$OLDPWD/gpurun_out/pmc_ivfflat.log" ); echo "exit $?" >> gpurun_out/pmc_ivfflat.log
      rm -f gpurun_out/pmc_ivfflat_summary.txt
      python tools/pmc_summary.py gpurun_out/pmc_ivfflat/r01_results.db gpurun_out/pmc_ivfflat_summary.txt '%k_list_scan%' '%k_select%' '%k_finalize%'
      rm -rf gpurun_out/pmc_ivfflat ;;
    variants)
      # cost split of k_pq_scan8 (unfiltered form): 0 = real kernel, 1 = gathers + ONE add, 2 = no LDS gather
      for v in 0 1 2; do RSX_SCAN8_VARIANT=$v timeout 600 python bench.py --steps 5 --warmup 2 --cpu-queries 0 --no-recall --no-configs --param pq_filter=0 > gpurun_out/bench_var$v.json 2> gpurun_out/bench_var$v.log; done ;;
    ab_lib)
      # same-box A/B of two builds: retrieval-scaling_amd/csrc/librsx_head.so (copied there by hand) vs the current one
      for r in 1 2; do
        RSX_LIB=$PWD/retrieval-scaling_amd/csrc/librsx_head.so timeout 600 python bench.py --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs > gpurun_out/bench_ab_head$r.json 2> gpurun_out/bench_ab_head$r.log
        timeout 600 python bench.py --steps 10 --warmup 3 --cpu-queries 0 --no-recall --no-configs > gpurun_out/bench_ab_new$r.json 2> gpurun_out/bench_ab_new$r.log
      done ;;
    bench_diag)
      timeout 900 python bench.py --diag --no-recall --cpu-queries 0 > gpurun_out/bench_diag.json 2> gpurun_out/bench_diag.log; echo "exit $?" >> gpurun_out/bench_diag.log ;;
    prof)
      ( cd /tmp && timeout 1700 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.log" ); echo "exit $?" >> gpurun_out/prof.log
      find gpurun_out/prof -name "*stats*" | head >> gpurun_out/prof.log ;;
    pmc)
      ( cd /tmp && timeout 1700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o r01 -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > "$OLDPWD/gpurun_out/pmc_bench.json" 2> "$OLDPWD/gpurun_out/pmc.log" ); echo "exit $?" >> gpurun_out/pmc.log ;;
  esac
done
ls -la gpurun_out > gpurun_out/ls.txt
for f in gpurun_out/*.log; do echo "== $f"; tail -n 5 "$f"; done
