#!/usr/bin/env python3
"""One line per bench JSON file: ms/step, the scan stage, the fixed part, queries/s.  usage: show_bench.py file.json [label]"""
import json
import sys

txt = open(sys.argv[1]).read().strip().splitlines()
lines = [l for l in txt if l.startswith("{")]
if not lines:
    print(f"{sys.argv[1]}: no JSON line")
    sys.exit(0)
j = json.loads(lines[-1])
s = j.get("stage_ms_per_step") or {}
label = sys.argv[2] if len(sys.argv) > 2 else j.get("config", {}).get("workload", "")
scan = s.get("scan", 0.0)
r = j.get("roofline") or {}
print(f"{label}: {j['ms_per_step']:.3f} ms/step, scan {scan:.3f}, stage total {s.get('total', 0.0):.3f}, fixed {j['ms_per_step'] - scan:.3f} "
      f"(stages {s.get('total', 0.0) - scan:.3f}), {j['value']:.0f} q/s, frac {r.get('frac')}, fallbacks {j.get('certificate_fallback_fraction')}")
print("   stages:", {k: v for k, v in s.items() if v})
oc = j.get("one_call_all_queries")
if oc:
    seq = oc.get("sequential", oc)      # (round-4 lines carried a sequential / pipelined pair)
    print(f"   one call of {oc['queries']} queries: {seq['ms']:.3f} ms ({seq['queries_per_s']:.0f} q/s)")
