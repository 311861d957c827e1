"""On-GPU diagnostic: where does the HIP synthetic generator deviate from the oracle's?"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd"))
import rsx
from oracle import oracle as o
np.set_printoptions(linewidth=200)
for (d, nc, sig, n) in [(8, 4, 0.5, 6), (8, 1, 0.0, 3), (8, 4, 0.0, 6), (8, 1, 0.5, 3), (768, 64, 0.5, 2000)]:
    g = rsx.synth_vectors(d, nc, 1234, 10000, sig, 0, n)
    c = o.synth_vectors(d, nc, 1234, 10000, sig, 0, n)
    neq = (g.view(np.uint16) != c.view(np.uint16))
    print(f"d={d} nc={nc} sigma={sig} n={n}: mismatches {neq.sum()} / {neq.size}; max abs diff {np.abs(g.astype(np.float32)-c.astype(np.float32)).max()}")
    if d == 8:
        print(" gpu", g[:3].tolist()); print(" cpu", c[:3].tolist())
    else:
        idx = np.argwhere(neq)[:8]
        for i, t in idx:
            print("  ", i, t, float(g[i, t]), float(c[i, t]), hex(g.view(np.uint16)[i, t]), hex(c.view(np.uint16)[i, t]))
