# round 3: the reference's own operating points on the engine as it stands
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python tools/exp_scan.py --rounds 1 --steps 5 --warmup 2 ${HEAD_SETS:---set k=10 --set k=100 --set k=1000 --set k=2000} > gpurun_out/ops_headline_k.jsonl 2> gpurun_out/ops_headline_k.log; echo "exit $?" >> gpurun_out/ops_headline_k.log
timeout 900 python tools/exp_scan.py --rounds 1 --steps 3 --warmup 1 --m 16 --nlist 8192 --nprobe 512 ${M16_SETS:---set k=10 --set k=100 --set k=1000} > gpurun_out/ops_m16.jsonl 2> gpurun_out/ops_m16.log; echo "exit $?" >> gpurun_out/ops_m16.log
for f in gpurun_out/ops_headline_k.jsonl gpurun_out/ops_m16.jsonl; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    r = json.loads(l); st = r["stages"]
    print(r["set"], "qps", r["qps"], "ms", r["ms_per_step"], "scan0", st["scan0"], "scan", st["scan"], "select", st["select"], "fin", st["finalize"], "total", st["total"],
          "fallbacks", r["fallback_queries"], "overflow", r["fallback_overflow"], "cand", r["cand_mean"], r["cand_max"])
PY
done
tail -n 2 gpurun_out/ops_headline_k.log gpurun_out/ops_m16.log
