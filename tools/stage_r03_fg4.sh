#!/bin/bash
# Flat GEMM prototype runs (tools/proto/flat_gemm4*): numerical check on 200k rows, then 10M-row timings -> gpurun_out/${TAG}_fg4.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}/tools/proto"; mkdir -p ../../gpurun_out
TAG=${TAG:-r03fg}
OUT=../../gpurun_out/${TAG}_fg4.txt; : > $OUT
for b in ${BINS:-flat_gemm4 flat_gemm4_s3}; do
  echo "== $b" >> $OUT
  timeout 120 ./$b 200192 1 >> $OUT 2>&1
  timeout 120 ./$b 10000000 >> $OUT 2>&1
done
if [ -n "$PMC_BIN" ]; then
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/pmc_fg -o fg -- $OLDPWD/$PMC_BIN 10000000 > /dev/null 2>&1 )
  python ../pmc_summary.py /tmp/pmc_fg/fg_results.db ../../gpurun_out/${TAG}_fg4_pmc.md '%k_fg%' >> $OUT 2>&1
  cat ../../gpurun_out/${TAG}_fg4_pmc.md >> $OUT
fi
cat $OUT
