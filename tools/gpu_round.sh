#!/bin/bash
# ONE parameterised GPU-box session script (replaces the per-experiment tools/stage_*.sh of rounds 1-3).
# Usage: [TAG=r04a] [PYTEST_ARGS=...] [PYTEST_K='expr with spaces'] tools/gpu_round.sh stage [stage ...]      (everything lands in gpurun_out/)
#   env        box facts (GPU, cores, RAM, import faiss)
#   tests      pytest -m gpu (stop at first failure; PYTEST_ARGS narrows it)     tests_all: without -x
#   smoke      __graft_entry__.smoke()
#   bench      the default bench line (all configs)                               bench_fast: headline only
#   prof       rocprofv3 --kernel-trace --stats of the headline bench  -> ${TAG}_rocprof_stats_ivfpq100M.md + ${TAG}_timeline.md (last batch)
#   pmc_fetch  FETCH_SIZE pass -> ${TAG}_pmc_fetch_size.md + stamped pmc_traffic.json
#   stamp      copy the pmc_fetch stage's pmc_traffic.json over profiles/pmc_traffic.json ON THE BOX, so that a later bench stage of the same call reads it
#              (final evidence session: env pmc_fetch stamp tests smoke bench prof per_rank latency flat; then copy gpurun_out/pmc_traffic.json to profiles/ here)
#   pmc_sq     two SQ counter passes of the scan kernel -> ${TAG}_pmc_sq_counters.md
#   per_rank   one rank of an N = 2 / 4 / 8 run and of config 5 on this one GPU -> ${TAG}_per_rank_workloads.txt
#   m16        the reference's shipped IVF-PQ point (M 16, nlist 8192, nprobe 512; k 10 and 1000)
#   m16_prof   rocprofv3 kernel stats of it              m16_pmc: SQ counters of its scan
#   largek     the headline index at k = 100 / 1000 / 2000
#   flat       Flat 10M (k 10 and 1000) + rocprof stats          ivfflat: IVF-Flat configs     ivfflat_prof: nlist 2048 / nprobe 128 + rocprof stats
#   latency    single-query latency protocol
#   cmd        run "$CMD" (free-form, logged to ${TAG}_cmd.log)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r06}
O=$PWD/gpurun_out
FAST="--cpu-queries 0 --no-recall --no-configs"
for s in "$@"; do
  case $s in
    env)
      { rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; python -c "import faiss" 2>&1 | tail -1; } > $O/${TAG}_gpu_box_env.txt 2>&1 ;;
    tests)
      timeout 1700 python -m pytest tests -q -m gpu -x --timeout 600 -p no:cacheprovider ${PYTEST_ARGS:-} ${PYTEST_K:+-k "$PYTEST_K"} > $O/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${TAG}_pytest_gpu.txt
      tail -n 15 $O/${TAG}_pytest_gpu.txt | cut -c1-220 ;;
    tests_all)
      timeout 1700 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider ${PYTEST_ARGS:-} ${PYTEST_K:+-k "$PYTEST_K"} > $O/${TAG}_pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/${TAG}_pytest_gpu.txt
      tail -n 15 $O/${TAG}_pytest_gpu.txt | cut -c1-220 ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> $O/${TAG}_smoke.log ;;
    bench)
      timeout 1500 python bench.py --steps ${STEPS:-20} --warmup 5 ${BENCH_ARGS:-} > $O/${TAG}_bench_ivfpq100M.json 2> $O/${TAG}_bench.log; echo "exit $?" >> $O/${TAG}_bench.log ;;
    bench_fast)
      timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 $FAST ${BENCH_ARGS:-} > $O/${TAG}_bench_fast.json 2> $O/${TAG}_bench_fast.log; echo "exit $?" >> $O/${TAG}_bench_fast.log
      python tools/show_bench.py $O/${TAG}_bench_fast.json ;;
    prof)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG -- python $OLDPWD/bench.py --steps 5 --warmup 2 $FAST ${BENCH_ARGS:-} > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.log ); echo "exit $?" >> $O/${TAG}_prof.log
      python tools/rocprof_summary.py $O/prof/${TAG}_results.db $O/${TAG}_rocprof_stats_ivfpq100M.md "IVF-PQ 100M x 768, M=96, nlist=4096, nprobe=32, batch=1024 (python bench.py --steps 5 --warmup 2 $FAST ${BENCH_ARGS:-})"
      python tools/timeline.py $O/prof/${TAG}_results.db $O/${TAG}_timeline.md > /dev/null
      rm -rf $O/prof ;;
    pmc_fetch)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o $TAG -- python $OLDPWD/bench.py --steps 2 --warmup 1 $FAST ${BENCH_ARGS:-} > /dev/null 2> $O/${TAG}_pmc.log ); echo "exit $?" >> $O/${TAG}_pmc.log
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o $TAG -- python $OLDPWD/bench.py --steps 2 --warmup 1 $FAST ${BENCH_ARGS:-} > /dev/null 2> $O/${TAG}_pmc_mfma.log ); echo "exit $?" >> $O/${TAG}_pmc_mfma.log
      rm -f $O/${TAG}_pmc_fetch_size.md
      python tools/pmc_summary.py /tmp/pmc_fetch/${TAG}_results.db $O/${TAG}_pmc_fetch_size.md '%k_pq_scan%' '%k_pq_prepass%' '%k_pq_rot%'
      python tools/pmc_summary.py /tmp/pmc_mfma/${TAG}_results.db $O/${TAG}_pmc_fetch_size.md '%k_pq_scan%'
      python tools/update_pmc_traffic.py /tmp/pmc_fetch/${TAG}_results.db $O/pmc_traffic.json 100000000 1 /tmp/pmc_mfma/${TAG}_results.db
      rm -rf /tmp/pmc_fetch /tmp/pmc_mfma ;;
    stamp)
      cp $O/pmc_traffic.json profiles/pmc_traffic.json ;;
    pmc_sq)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/pmc_sq -o $TAG -- python $OLDPWD/bench.py --steps 2 --warmup 1 $FAST ${BENCH_ARGS:-} > /dev/null 2> $O/${TAG}_pmc_sq.log ); echo "exit $?" >> $O/${TAG}_pmc_sq.log
      rm -f $O/${TAG}_pmc_sq_counters.md
      python tools/pmc_summary.py $O/pmc_sq/${TAG}_results.db $O/${TAG}_pmc_sq_counters.md '%k_pq_scan%' '%k_pq_rot%' '%k_finalize%' '%k_pq_lut%'
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq2 -o $TAG -- python $OLDPWD/bench.py --steps 2 --warmup 1 $FAST ${BENCH_ARGS:-} > /dev/null 2> $O/${TAG}_pmc_sq2.log ); echo "exit $?" >> $O/${TAG}_pmc_sq2.log
      python tools/pmc_summary.py $O/pmc_sq2/${TAG}_results.db $O/${TAG}_pmc_sq_counters.md '%k_pq_scan%'
      rm -rf $O/pmc_sq $O/pmc_sq2 ;;
    per_rank)
      : > $O/${TAG}_per_rank_workloads.txt
      for n in 50000000 25000000 12500000 125000000; do
        timeout 600 python bench.py --n $n --steps 20 --warmup 5 $FAST ${BENCH_ARGS:-} > $O/${TAG}_n$n.json 2> $O/${TAG}_n$n.log
        python tools/show_bench.py $O/${TAG}_n$n.json "n=$n batch=1024" >> $O/${TAG}_per_rank_workloads.txt
      done
      cat $O/${TAG}_per_rank_workloads.txt ;;
    m16)
      timeout 900 python tools/bench_configs.py ivfpq_ref --steps ${STEPS:-5} ${M16_ARGS:-} > $O/${TAG}_ivfpq_m16.json 2> $O/${TAG}_ivfpq_m16.log; echo "exit $?" >> $O/${TAG}_ivfpq_m16.log
      cut -c1-2500 $O/${TAG}_ivfpq_m16.json; tail -n 3 $O/${TAG}_ivfpq_m16.log | cut -c1-300 ;;
    m16_prof)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof_m16 -o $TAG -- python $OLDPWD/tools/bench_configs.py ivfpq_ref --steps 3 --check 0 ${M16_ARGS:-} > /dev/null 2> $O/${TAG}_prof_m16.log ); echo "exit $?" >> $O/${TAG}_prof_m16.log
      python tools/rocprof_summary.py $O/prof_m16/${TAG}_results.db $O/${TAG}_rocprof_stats_ivfpq_m16.md "IVF-PQ 100M x 768, M=16, nlist=8192, nprobe=512, batch=1024, k = 10 and 1000 (tools/bench_configs.py ivfpq_ref --steps 3 --check 0 ${M16_ARGS:-})"
      rm -rf $O/prof_m16; head -n 40 $O/${TAG}_rocprof_stats_ivfpq_m16.md | cut -c1-200 ;;
    m16_pmc)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/pmc_m16 -o $TAG -- python $OLDPWD/tools/bench_configs.py ivfpq_ref --steps 2 --check 0 --ks ${M16_PMC_K:-10} ${M16_ARGS:-} > /dev/null 2> $O/${TAG}_pmc_m16.log ); echo "exit $?" >> $O/${TAG}_pmc_m16.log
      rm -f $O/${TAG}_pmc_sq_m16.md
      python tools/pmc_summary.py $O/pmc_m16/${TAG}_results.db $O/${TAG}_pmc_sq_m16.md '%k_pq_scan%'
      rm -rf $O/pmc_m16 ;;
    largek)
      timeout 900 python tools/bench_configs.py largek --steps ${STEPS:-5} ${LARGEK_ARGS:-} > $O/${TAG}_largek.json 2> $O/${TAG}_largek.log; echo "exit $?" >> $O/${TAG}_largek.log
      cut -c1-2500 $O/${TAG}_largek.json; tail -n 3 $O/${TAG}_largek.log | cut -c1-300 ;;
    flat)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof_flat -o $TAG -- python $OLDPWD/tools/bench_configs.py flat --check 64 --steps 3 --ks 1000 > $O/${TAG}_flat10M.json 2> $O/${TAG}_flat10M.log ); echo "exit $?" >> $O/${TAG}_flat10M.log
      python tools/rocprof_summary.py $O/prof_flat/${TAG}_results.db $O/${TAG}_rocprof_stats_flat10M.md "Flat 10M x 768 batch 1024, k = 10 and 1000, + the small-batch searches (tools/bench_configs.py flat --check 64 --steps 3 --ks 1000)"
      rm -rf $O/prof_flat ;;
    ivfflat_prof)
      ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats -d $O/prof_ivf -o $TAG -- python $OLDPWD/tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 --ks 1000 > $O/${TAG}_ivfflat20M_nprobe128.json 2> $O/${TAG}_ivfflat_prof.log ); echo "exit $?" >> $O/${TAG}_ivfflat_prof.log
      python tools/rocprof_summary.py $O/prof_ivf/${TAG}_results.db $O/${TAG}_rocprof_stats_ivfflat20M_nprobe128.md "IVF-Flat 20M x 768, nlist 2048, nprobe 128, batch 1024, k = 10 and 1000 (tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 2 --steps 3 --ks 1000)"
      rm -rf $O/prof_ivf ;;
    ivfflat)
      timeout 900 python tools/bench_configs.py ivfflat ${IVFFLAT_ARGS:-} > $O/${TAG}_ivfflat.json 2> $O/${TAG}_ivfflat.log; echo "exit $?" >> $O/${TAG}_ivfflat.log
      cut -c1-1500 $O/${TAG}_ivfflat.json ;;
    latency)
      timeout 500 python tools/bench_configs.py latency > $O/${TAG}_latency_ivfpq100M.json 2> $O/${TAG}_latency.log; echo "exit $?" >> $O/${TAG}_latency.log ;;
    cmd)
      timeout ${CMD_TIMEOUT:-900} bash -c "$CMD" > $O/${TAG}_cmd.log 2>&1; echo "exit $?" >> $O/${TAG}_cmd.log; tail -n ${CMD_TAIL:-40} $O/${TAG}_cmd.log | cut -c1-400 ;;
    *) echo "unknown stage $s" ;;
  esac
done
ls -la gpurun_out | tail -n 12
