#!/usr/bin/env python3
"""Batch pipeline (rsx_api.hip: search_impl) on the headline index: one search call of 4096 queries, sequential against pipelined
for a sweep of pipeline_reserve (CUs the scan grid leaves free).  usage: pipe_sweep.py [--n N] [--m M --nlist L --nprobe P]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import torch, rsx
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000_000); ap.add_argument("--m", type=int, default=96)
ap.add_argument("--nlist", type=int, default=4096); ap.add_argument("--nprobe", type=int, default=32)
ap.add_argument("--nq", type=int, default=4096); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reserves", default="0,8,16,32,64")
a = ap.parse_args()
D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999
dev = torch.device("cuda", 0)
ix = rsx.IndexIVFPQ(None, D, a.nlist, a.m, 8, rsx.METRIC_INNER_PRODUCT)
nt = min(a.n, 256 * a.nlist)
xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
rsx.synth_vectors(D, NC, SC, SX, 0.5, 0, nt, out=xt)
ix.train(xt); del xt
ix.nprobe = a.nprobe
buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
for c0 in range(0, a.n, buf.shape[0]):
    nb = min(buf.shape[0], a.n - c0)
    rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb]); ix.add(buf[:nb])
del buf
Q = torch.empty((a.nq, D), dtype=torch.float16, device=dev)
rsx.synth_queries(D, NC, SC, SX, 0.5, a.n, SQ, 0.1, 0, a.nq, out=Q)

def run(reps=5):
    ix.search(Q, a.k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ix.search(Q, a.k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

ix.set_param("pipeline", 0)
ms0, ref = run()
print(json.dumps({"mode": "sequential", "ms": round(ms0, 3), "qps": round(a.nq / ms0 * 1e3)}), flush=True)
ix.set_param("pipeline", 1)
for r in [int(t) for t in a.reserves.split(",")]:
    ix.set_param("pipeline_reserve", r)
    ms, out = run()
    print(json.dumps({"mode": "pipelined", "reserve": r, "ms": round(ms, 3), "qps": round(a.nq / ms * 1e3),
                      "same": bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))}), flush=True)
ix.set_param("pipeline", 0)
ms0, _ = run()
print(json.dumps({"mode": "sequential again", "ms": round(ms0, 3)}), flush=True)
