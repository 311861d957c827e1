#!/usr/bin/env python3
"""Phase trace of k_pq_final_tab (needs RSX_LIB=.../librsx_measure.so): microseconds between the marks of thread 0, averaged over the
first 256 workgroups (= queries) of the last launch.  usage: exp_ft_trace.py [n] [k ...]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch, rsx
import bench_dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ks = [int(x) for x in sys.argv[2:]] or [1000]
ix = bench_dist.standard_index(n)
Q = torch.empty((4 * 1024, 768), dtype=torch.float16, device="cuda")
rsx.synth_queries(768, 4096, 1234, 10000, 0.5, n, 999, 0.1, 0, 4 * 1024, out=Q)
names = ["table", "re-score", "threshold (radix)", "collect", "sort", "emit"]
for k in ks:
    for i in range(3): ix.search(Q[i * 1024:(i + 1) * 1024], k)
    ix.set_param("profile", 1)
    ix.search(Q[3 * 1024:], k)
    torch.cuda.synchronize()
    tr = np.zeros((256, 8), dtype=np.uint64)
    assert rsx.lib().rsx_debug_ft_trace(tr.ctypes.data_as(ctypes.c_void_p)) == 0
    t = tr.astype(np.int64)
    d = [(t[:, i + 1] - t[:, i]).mean() / 100.0 for i in range(6)]
    print(f"k={k}: finalize stage {ix.get_timing('finalize'):.3f} ms; per workgroup (us): " + "  ".join(f"{nm} {v:.1f}" for nm, v in zip(names, d)) +
          f"  | total {(t[:, 6] - t[:, 0]).mean() / 100.0:.1f} us; span of the 256 {(t[:, 6].max() - t[:, 0].min()) / 100.0:.1f} us")
    if k <= 25:      # the K' path: gather + select, then k_finalize
        tr = np.zeros((256, 8), dtype=np.uint64)
        assert rsx.lib().rsx_debug_gs_trace(tr.ctypes.data_as(ctypes.c_void_p)) == 0
        t = tr.astype(np.int64)
        gn = ["descriptors (3 dependent loads)", "scan", "copy keys", "select K'", "store"]
        d = [(t[:, i + 1] - t[:, i]).mean() / 100.0 for i in range(5)]
        print(f"k={k}: select stage {ix.get_timing('select'):.3f} ms; k_pq_gather_select per workgroup (us): " + "  ".join(f"{nm} {v:.1f}" for nm, v in zip(gn, d)) +
              f"  | total {(t[:, 5] - t[:, 0]).mean() / 100.0:.1f} us; span of the 256 {(t[:, 5].max() - t[:, 0].min()) / 100.0:.1f} us")
    if k > 25:
        tr2 = np.zeros((256, 8), dtype=np.uint64)
        assert rsx.lib().rsx_debug_ft_trace2(tr2.ctypes.data_as(ctypes.c_void_p)) == 0
        t2 = tr2.astype(np.int64)
        parts = [("count", t2[:, 0] - t[:, 1]), ("k-th approximate + bitmap", t2[:, 1] - t2[:, 0]), ("stage 1", t2[:, 2] - t2[:, 1]),
                 ("k-th exact of stage 1", t2[:, 3] - t2[:, 2]), ("stage 2", t2[:, 4] - t2[:, 3]), ("to the end of the phase", t[:, 2] - t2[:, 4])]
        print(f"k={k}: the re-score phase in parts (us): " + "  ".join(f"{nm} {v.mean() / 100.0:.1f}" for nm, v in parts))
    ix.set_param("profile", 2)
    ix.search(Q[3 * 1024:], k)
    print(f"k={k}: candidate keys per query {ix.get_timing('cand_keys') / 1024:.0f} (max {ix.get_timing('cand_keys_max'):.0f}); re-scored exactly by k_pq_final_tab per query {ix.get_timing('final_tab_rescored') / 1024:.0f}")
    ix.set_param("profile", 0)
