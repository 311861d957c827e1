import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np
import rsx as gpu
from oracle import oracle as orc
d, M, nlist, n = 64, 16, 8, 5000
x = orc.synth_vectors(d, nlist, 71, 72, 0.5, 0, n)
x[100:140] = x[7]
q = np.concatenate([x[7:8], orc.synth_queries(d, nlist, 71, 72, 0.5, n, 73, 0.1, 0, 9)], 0)
x32, q32 = x.astype(np.float32), q.astype(np.float32)
cen = orc.kmeans(0, x32, nlist, 4, 1234)
a, _ = orc.assign_ip(cen, x32)
cb = orc.pq_train(orc.residuals(cen, x32, a)[:2000], M, 3, 1234)
lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, orc.residuals(cen, x32, a)), nlist)
print("list sizes", np.bincount(a, minlength=nlist))
for layout in (1, 0):
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    ix.set_param("pq_layout", layout)
    ix.set_centroids(cen); ix.set_codebooks(cb); ix.add(x)
    ix.nprobe = nlist
    for k in (64, 500, 1000, 2048):
        for knobs in ({}, {"pq_gather": 0}, {"pq_prepass_fused": 0}, {"pq_filter": 0}, {"pq_fast": 0}):
            for kk, vv in knobs.items(): ix.set_param(kk, vv)
            ix.set_param("profile", 1)
            D, I = ix.search(q, k)
            fb, sc = ix.get_timing("fallback_queries"), ix.get_timing("second_chance_queries")
            ix.set_param("profile", 0)
            for kk in knobs: ix.set_param(kk, 1)
            Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, nlist, k)
            badq = [int(i) for i in range(len(q)) if not (np.array_equal(D[i], Dr[i]) and np.array_equal(I[i], Ir[i]))]
            msg = ""
            if badq:
                i = badq[0]; pos = np.nonzero((D[i] != Dr[i]) | (I[i] != Ir[i]))[0]
                miss = np.setdiff1d(Ir[i], I[i]); extra = np.setdiff1d(I[i], Ir[i])
                msg = f" first bad q{i}: {len(pos)} positions from {pos[:3]}, missing ids {miss[:5]} (lists {a[miss[:5]]}, rows-in-list n/a) extra {extra[:5]}"
            print(f"layout={layout} k={k} knobs={knobs}: bad queries {badq} fb={fb} second={sc}{msg}", flush=True)
