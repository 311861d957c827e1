#!/bin/bash
# Round-5 GPU session 23: PMC counters of the IVF-Flat scans at nlist 2048 / nprobe 128: k_list_scan3 (128 queries per group) against the
# 64-query form of k_list_scan2 — FETCH_SIZE (HBM bytes), TCC requests / hits / misses, TCP latency and stall
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${TAG:-r05v}
rm -f $O/${T}_pmc_list_scans.md
pmc_run() {   # $1 = tag, $2 = counters, rest = bench_configs args
  local tag=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$tag
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$tag -o $tag -- python $R/tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 0 --steps 2 "$@" > /dev/null 2> $O/${T}_${tag}.log ); echo "exit $?" >> $O/${T}_${tag}.log
  echo "## $tag: $ctr ($*)" >> $O/${T}_pmc_list_scans.md
  python tools/pmc_summary.py /tmp/pmc_$tag/${tag}_results.db $O/${T}_pmc_list_scans.md '%k_list_scan3%' '%k_list_scan2<true%'
  rm -rf /tmp/pmc_$tag
}
pmc_run s3_fetch FETCH_SIZE
pmc_run s2_fetch FETCH_SIZE --param ivf_qtiles=4
pmc_run s3_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pmc_run s2_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" --param ivf_qtiles=4
pmc_run s3_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE"
pmc_run s2_tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" --param ivf_qtiles=4
pmc_run s3_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS"
cat $O/${T}_pmc_list_scans.md | cut -c1-200
