#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
RSX_LIB=$PWD/retrieval-scaling_amd/csrc/librsx_measure.so timeout 300 python tools/exp_pp4_trace.py > gpurun_out/${TAG:-r03pp4}_trace.txt 2>&1
tail -5 gpurun_out/${TAG:-r03pp4}_trace.txt
