#!/usr/bin/env python3
"""Diagnostics: the reference's M = 16 / nlist 8192 / nprobe 512 point — why does k_pq_final_tab flag queries?"""
import os, sys, time, json
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import torch, rsx
D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999
n = int(os.environ.get("N", 100_000_000)); M, nlist, nprobe, nq = 16, 8192, 512, 1024
dev = torch.device("cuda", 0)
ix = rsx.IndexIVFPQ(None, D, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
nt = min(n, 256 * nlist); xt = torch.empty((nt, D), dtype=torch.float16, device=dev); stride = max(1, n // nt)
for b in range(0, nt, 4096):
    nb = min(4096, nt - b); rsx.synth_vectors(D, NC, SC, SX, 0.5, (b * stride) % max(1, n - nb), nb, out=xt[b:b + nb])
ix.train(xt); del xt; ix.nprobe = nprobe
buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
for c0 in range(0, n, buf.shape[0]):
    nb = min(buf.shape[0], n - c0); rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb]); ix.add(buf[:nb])
del buf
Q = torch.empty((nq, D), dtype=torch.float16, device=dev); rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, nq, out=Q)
res = {}
for k in (10, 1000):
    for tab in (0, 1):
        ix.set_param("pq_final_tab", tab); ix.set_param("profile", 1)
        Dg, Ig = ix.search(Q, k)
        Ih = Ig.cpu().numpy(); Dh = Dg.cpu().numpy()
        dup = sum(len(set(r.tolist())) != k for r in Ih)
        # exact score ties among each query's returned top k (adjacent equal scores)
        ties = int((Dh[:, 1:] == Dh[:, :-1]).sum())
        res[f"k{k}_tab{tab}"] = {"fallback": ix.get_timing("fallback_queries"), "overflow": ix.get_timing("fallback_overflow_queries"),
                                "tie_flagged": ix.get_timing("fallback_tie_queries"), "tie_max": ix.get_timing("fallback_tie_max"),
                                "second": ix.get_timing("second_chance_queries"), "queries_with_duplicate_ids": dup, "adjacent_equal_scores": ties}
        if tab == 0: I0, D0 = Ih, Dh
        else: res[f"k{k}_same_as_old"] = bool(np.array_equal(I0, Ih) and np.array_equal(D0, Dh))
    ix.set_param("scan_kernel", 2); De, Ie = ix.search(Q[:32], k); ix.set_param("scan_kernel", 0)
    res[f"k{k}_first32_equal_exact_kernel"] = bool(np.array_equal(Ie.cpu().numpy(), I0[:32]) and np.array_equal(De.cpu().numpy(), D0[:32]))
print(json.dumps(res, indent=1))
