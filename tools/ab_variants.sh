#!/bin/bash
# A/B of engine builds on the headline index: tools/ab_variants.sh <tag> ...  (tag "main" = librsx.so, else librsx_<tag>.so from tools/build_variant.sh);
# EXP_ARGS passes arguments to tools/exp_scan.py (default: the sliced layout, one round)
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "$@"; do
  lib=$PWD/retrieval-scaling_amd/csrc/librsx_$v.so; [ "$v" = main ] && lib=$PWD/retrieval-scaling_amd/csrc/librsx.so
  RSX_LIB=$lib timeout 600 python tools/exp_scan.py ${EXP_ARGS:---layouts 2 --rounds 1} > gpurun_out/ab_$v.txt 2> gpurun_out/ab_$v.log
  python - <<PY
import json
for l in open("gpurun_out/ab_$v.txt"):
    r = json.loads(l); print("$v", r.get("layout"), r["set"], "scan", r["stages"]["scan"], "ms/step", r["ms_per_step"], "fb", r["fallback_queries"], "cand", r["cand_mean"], "same", r["same_as_first"])
PY
done
