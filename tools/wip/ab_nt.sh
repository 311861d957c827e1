#!/bin/bash
# A/B of the nt cache policy on the streamed operands (one box): librsx.so against librsx_nt_{pq,list,flat}.so (csrc/Makefile).
# usage (on the GPU box): tools/wip/ab_nt.sh > gpurun_out/<tag>_ab_nt.txt
C=retrieval-scaling_amd/csrc
pick() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
for line in open(path):
    line = line.strip()
    if not line.startswith("{"): continue
    r = json.loads(line)
    if "by_k" in r:
        print(tag, r["config"], {k: (v["ms_per_step"], v["stage_ms"]["scan"]) for k, v in r["by_k"].items()})
    elif "stage_ms_per_step" in r:
        print(tag, "headline ms/step", r["ms_per_step"], "scan", r["stage_ms_per_step"].get("scan"), "q/s", r["value"])
    else:
        print(tag, r.get("config"), "ms/step", r.get("ms_per_step"), "scan", r.get("scan_ms"), r.get("roofline"))
PY
}
for rep in 1 2; do
for v in base pq; do
  L=$C/librsx.so; [ $v = base ] || L=$C/librsx_nt_$v.so
  RSX_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --cpu-queries 0 --no-recall --no-configs > /tmp/ab.json 2> /tmp/ab.log; pick "$v/$rep" /tmp/ab.json
done
done
for v in base pq; do
  L=$C/librsx.so; [ $v = base ] || L=$C/librsx_nt_$v.so
  RSX_LIB=$L timeout 600 python tools/bench_configs.py ivfpq_ref --steps 5 --check 0 --ks 10 > /tmp/ab.json 2> /tmp/ab.log; pick "$v" /tmp/ab.json
done
for v in base list base; do
  L=$C/librsx.so; [ $v = base ] || L=$C/librsx_nt_$v.so
  RSX_LIB=$L timeout 600 python tools/bench_configs.py ivfflat --nlist 2048 --nprobe 128 --check 0 > /tmp/ab.json 2> /tmp/ab.log; pick "$v" /tmp/ab.json
done
for v in base list; do
  L=$C/librsx.so; [ $v = base ] || L=$C/librsx_nt_$v.so
  RSX_LIB=$L timeout 600 python tools/bench_configs.py ivfflat --check 0 > /tmp/ab.json 2> /tmp/ab.log; pick "$v" /tmp/ab.json
done
for v in base flat base; do
  L=$C/librsx.so; [ $v = base ] || L=$C/librsx_nt_$v.so
  RSX_LIB=$L timeout 600 python tools/bench_configs.py flat --check 0 > /tmp/ab.json 2> /tmp/ab.log; pick "$v" /tmp/ab.json
done
