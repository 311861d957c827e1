#!/usr/bin/env python3
"""IVF-Flat 20M, nlist 2048, nprobe 128, k = 1000: stage timings and candidate counts of the staged filter."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import torch, rsx
D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999
n, nlist, nprobe, nq, k = 20_000_000, 2048, 128, 1024, 1000
dev = torch.device("cuda", 0)
ix = rsx.IndexIVFFlat(None, D, nlist, rsx.METRIC_INNER_PRODUCT)
xt = torch.empty((256 * nlist, D), dtype=torch.float16, device=dev); rsx.synth_vectors(D, NC, SC, SX, 0.5, 0, xt.shape[0], out=xt)
ix.train(xt); del xt; ix.nprobe = nprobe
buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
for c0 in range(0, n, buf.shape[0]):
    rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf); ix.add(buf)
del buf
Q = torch.empty((nq * 2, D), dtype=torch.float16, device=dev); rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, nq * 2, out=Q)
keys = ("coarse", "select_probe", "group", "scan0", "select0", "scan", "select", "finalize", "total", "stage_cand_keys", "cand_keys", "cand_keys_max", "fallback_queries")
for stages, pl in ((1, 0), (0, 0), (0, 2), (2, 2), (3, 1)):
    ix.set_param("ivf_stages", stages); ix.set_param("ivf_pre_lists", pl)
    ix.search(Q[:nq], k)
    for prof in (1, 2):
        ix.set_param("profile", prof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.search(Q[nq:], k)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        print(f"stages={stages} pre_lists={pl} profile={prof}: {ms:.2f} ms", {x: round(ix.get_timing(x), 3) for x in keys if ix.get_timing(x)}, flush=True)
    ix.set_param("profile", 0)
