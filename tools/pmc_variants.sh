export TMPDIR=/tmp RSX_PQ_LAYOUT=2
for v in "$@"; do
  lib=$PWD/retrieval-scaling_amd/csrc/librsx_$v.so; [ "$v" = main ] && lib=$PWD/retrieval-scaling_amd/csrc/librsx.so
  ( cd /tmp && RSX_LIB=$lib timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$v -o x -- python $OLDPWD/bench.py --steps 2 --warmup 1 --cpu-queries 0 --no-recall --no-configs > /dev/null 2> $OLDPWD/gpurun_out/pmc_$v.log )
  python tools/pmc_summary.py /tmp/pmc_$v/x_results.db gpurun_out/pmc_$v.md '%k_pq_scan%' > /dev/null 2>&1
  echo "$v: $(grep k_pq_scan gpurun_out/pmc_$v.md | head -2)"
done
