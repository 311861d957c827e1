#!/usr/bin/env python3
"""Phase trace of k_pq_prepass4 (needs RSX_LIB=.../librsx_measure.so): clk between the marks, averaged over 64 workgroups."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np
import torch, rsx
D, NC, n = 768, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
ix = rsx.IndexIVFPQ(rsx.IndexFlatIP(D), D, 4096, 96, 8, rsx.METRIC_INNER_PRODUCT, device=0)
nt = min(n, 256 * 4096); xt = torch.empty((nt, D), dtype=torch.float16, device=dev); stride = max(1, n // nt)
for b in range(0, nt, 4096):
    nb = min(4096, nt - b); rsx.synth_vectors(D, NC, 1234, 10000, 0.5, (b * stride) % max(1, n - nb), nb, out=xt[b:b + nb])
ix.train(xt); del xt
buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
for c0 in range(0, n, 1_000_000):
    nb = min(1_000_000, n - c0); rsx.synth_vectors(D, NC, 1234, 10000, 0.5, c0, nb, out=buf[:nb]); ix.add(buf[:nb])
del buf; ix.nprobe = 32
Q = torch.empty((4 * 1024, D), dtype=torch.float16, device=dev)
rsx.synth_queries(D, NC, 1234, 10000, 0.5, n, 999, 0.1, 0, 4 * 1024, out=Q)
for i in range(4): ix.search(Q[i * 1024:(i + 1) * 1024], 10)
torch.cuda.synchronize()
tr = np.zeros((64, 16, 8), dtype=np.uint64)
assert rsx.lib().rsx_debug_pp4_trace(tr.ctypes.data_as(ctypes.c_void_p)) == 0
t = tr.astype(np.int64)
names = ["setup+staging issue", "barrier 1", "scan", "barrier 2", "radix", "tail"]
for w in (0, 3, 15):
    d = [(t[:, w, i + 1] - t[:, w, i]).mean() for i in range(6)]
    print(f"wave {w:2d}: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d)) + f"  | total {(t[:, w, 6] - t[:, w, 0]).mean():.0f} (s_memtime ticks)")
