#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}/tools/proto"; mkdir -p ../../gpurun_out
TAG=${TAG:-r03fg}; OUT=../../gpurun_out/${TAG}_fg4.txt; : > $OUT
for b in ${BINS}; do echo "== $b" >> $OUT; timeout 120 ./$b ${NV:-10000000} >> $OUT 2>&1; done
cat $OUT
