#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}/tools/proto"; mkdir -p ../../gpurun_out
TAG=${TAG:-r03fg}; OUT=../../gpurun_out/${TAG}_fg4.txt; : > $OUT
echo "== flat_gemm4 pattern data" >> $OUT; timeout 200 ./flat_gemm4 10000000 0 0 >> $OUT 2>&1
echo "== flat_gemm4 random data" >> $OUT; timeout 300 ./flat_gemm4 10000000 0 1 >> $OUT 2>&1
cat $OUT
