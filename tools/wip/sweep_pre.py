#!/usr/bin/env python3
"""Sweep the threshold sample of the large-k pre-pass on the headline index (one build, several settings)."""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import torch, rsx
D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999
n, M, nlist, nprobe, nq = 100_000_000, 96, 4096, 32, 1024
dev = torch.device("cuda", 0)
ix = rsx.IndexIVFPQ(None, D, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
nt = 256 * nlist; xt = torch.empty((nt, D), dtype=torch.float16, device=dev); stride = n // nt
for b in range(0, nt, 4096):
    nb = min(4096, nt - b); rsx.synth_vectors(D, NC, SC, SX, 0.5, (b * stride) % (n - nb), nb, out=xt[b:b + nb])
ix.train(xt); del xt; ix.nprobe = nprobe
buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
for c0 in range(0, n, buf.shape[0]):
    rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf); ix.add(buf)
del buf
Q = torch.empty((nq * 6, D), dtype=torch.float16, device=dev); rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, nq * 6, out=Q)
ref = {}
for k in (100, 1000, 2000):
  for pre4 in (2, 1):
    for mult, mx in ((160, 32768), (160, 16384), (160, 8192), (16, 4096)):
        ix.set_param("pq_prepass4", pre4); ix.set_param("pq_pre_mult", mult); ix.set_param("pq_pre_max", mx)
        ix.search(Q[:nq], k)
        ix.set_param("profile", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(1, 6): Dq, Iq = ix.search(Q[s * nq:(s + 1) * nq], k)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
        st = {x: round(ix.get_timing(x) / 5, 3) for x in ("scan0", "scan", "select", "finalize", "total")}
        fb = ix.get_timing("fallback_queries") / 5
        ix.set_param("profile", 2); ix.search(Q[:nq], k); cand = ix.get_timing("cand_keys") / nq; ix.set_param("profile", 0)
        key = (Dq.cpu().numpy().tobytes(), Iq.cpu().numpy().tobytes())
        same = ref.setdefault(k, key) == key
        print(json.dumps({"k": k, "pq_prepass4": pre4, "mult": mult, "max": mx, "ms": round(el * 1e3, 3), "stages": st, "fallbacks": fb, "cand_mean": round(cand), "same_results": same}), flush=True)
