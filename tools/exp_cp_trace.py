#!/usr/bin/env python3
"""Phase trace of k_coarse_pick (needs RSX_LIB=.../librsx_measure.so): microseconds between the marks of thread 0, averaged over the first
256 workgroups (= queries) of the last launch.  usage: exp_cp_trace.py [n]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch, rsx
import bench_dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
ix = bench_dist.standard_index(n)
Q = torch.empty((4 * 1024, 768), dtype=torch.float16, device="cuda")
rsx.synth_queries(768, 4096, 1234, 10000, 0.5, n, 999, 0.1, 0, 4 * 1024, out=Q)
names = ["query + row load", "k-th key (32 steps)", "candidates", "exact chains", "sort", "probes"]
for i in range(3): ix.search(Q[i * 1024:(i + 1) * 1024], 10)
ix.set_param("profile", 1)
ix.search(Q[3 * 1024:], 10)
torch.cuda.synchronize()
tr = np.zeros((256, 8), dtype=np.uint64)
assert rsx.lib().rsx_debug_cp_trace(tr.ctypes.data_as(ctypes.c_void_p)) == 0
t = tr.astype(np.int64)
d = [(t[:, i + 1] - t[:, i]).mean() / 100.0 for i in range(6)]
print(f"select_probe stage {ix.get_timing('select_probe'):.3f} ms; k_coarse_pick per workgroup (us): " + "  ".join(f"{nm} {v:.1f}" for nm, v in zip(names, d)) +
      f"  | total {(t[:, 6] - t[:, 0]).mean() / 100.0:.1f} us; span of the 256 {(t[:, 6].max() - t[:, 0].min()) / 100.0:.1f} us")
