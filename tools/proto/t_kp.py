import sys, numpy as np
sys.path.insert(0, '/root/repo/retrieval-scaling_amd'); sys.path.insert(0, '/root/repo')
import rsx
d, n, nlist, M, nq, k = 768, 400000, 16, 96, 256, 10
x = rsx.synth_vectors(d, 16, 1234, 10000, 0.5, 0, n)
q = rsx.synth_queries(d, 16, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
ix = rsx.IndexIVFPQ(None, d, nlist, M, 8, rsx.METRIC_INNER_PRODUCT, device=0)
ix.train(x[:20000]); ix.add(x); ix.nprobe = 8
ix.set_param("scan_kernel", 2); De, Ie = ix.search(q, k); ix.set_param("scan_kernel", 0)
for kp in (0, 128, 512, 1024, 2048):
    for chunk in (0, 8192):
        ix.set_param("pq_fast_kp", kp); ix.set_param("scan_chunk", chunk); ix.set_param("profile", 2)
        D, I = ix.search(q, k)
        bad = np.nonzero((I != Ie).any(1) | (D != De).any(1))[0]
        print(f"kp={kp} chunk={chunk}: differing {len(bad)} fallbacks {ix.get_timing('fallback_queries')} survivors mean {ix.get_timing('cand_keys')/nq:.0f} max {ix.get_timing('cand_keys_max'):.0f}", bad[:8])
print("---- isolate")
ix.set_param("scan_chunk", 0); ix.set_param("pq_fast_kp", 512); ix.set_param("profile", 2)
D, I = ix.search(q, k)
bad = np.nonzero((I != Ie).any(1) | (D != De).any(1))[0]
print("bad", bad)
sub = q[bad]
D2, I2 = ix.search(sub, k)
print("bad ones as their own batch: differing", int(((I2 != Ie[bad]).any(1) | (D2 != De[bad]).any(1)).sum()), "fallbacks", ix.get_timing('fallback_queries'))
ix.set_param("pq_filter", 0)
D3, I3 = ix.search(q, k)
print("pq_filter=0: differing", int(((I3 != Ie).any(1) | (D3 != De).any(1)).sum()), "fallbacks", ix.get_timing('fallback_queries'))
ix.set_param("pq_filter", 1)
ix.set_param("pq_layout", 0) if False else None
for b in bad[:3]:
    print("q", b, "exact", Ie[b].tolist(), "got", I[b].tolist())
