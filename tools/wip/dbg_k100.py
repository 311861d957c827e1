#!/usr/bin/env python3
"""Per-call host times around a change of k on one index (bench.py's reference_n_docs leg showed 14 ms per step at k = 100 against 3.7 ms of stages)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch, rsx, bench_dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ix = bench_dist.standard_index(n)
nq = 1024
Q = torch.empty((13 * nq, 768), dtype=torch.float16, device="cuda")
rsx.synth_queries(768, 4096, 1234, 10000, 0.5, n, 999, 0.1, 0, 13 * nq, out=Q)
def run(k, idx, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.search(Q[idx * nq:(idx + 1) * nq], k)
    torch.cuda.synchronize(); print(f"{label} k={k} batch {idx}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
for i in range(4): run(10, i, "warm")
for kk in (100, 1000, 100):
    for i in range(2): run(kk, i, "untimed")
    ix.set_param("profile", 1)
    for i in range(3, 8): run(kk, i, "timed")
    print({s: round(ix.get_timing(s) / 5, 4) for s in ("scan0", "scan", "select", "finalize", "total")}, "fb", ix.get_timing("fallback_queries"), "2nd", ix.get_timing("second_chance_queries"))
    ix.set_param("profile", 0)
