#!/usr/bin/env python3
"""Diagnostic: how wide is the 8-bit table's rigorous error bound (eps) against the score distribution of the vectors a query
scans, and how many candidates would a threshold tau = s_k - c*eps admit, for several k?  (numpy restatement on one index)"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np


def main():
    import torch, rsx
    from oracle import oracle as orc
    D, NC, n, nlist, M, nprobe = 768, 4096, int(os.environ.get("N", 20_000_000)), 4096, int(os.environ.get("M", 96)), 32
    dev = torch.device("cuda", 0)
    ix = rsx.IndexIVFPQ(None, D, nlist, M, 8, 0)
    nt = 256 * nlist
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    rsx.synth_vectors(D, NC, 1234, 10000, 0.5, 0, nt, out=xt)
    ix.train(xt); del xt
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, n, 1_000_000):
        rsx.synth_vectors(D, NC, 1234, 10000, 0.5, c0, 1_000_000, out=buf); ix.add(buf)
    q = rsx.synth_queries(D, NC, 1234, 10000, 0.5, n, 999, 0.1, 0, 8).astype(np.float32)
    cen, cb = ix.get_centroids(), ix.get_codebooks()
    pid, pdis = orc.coarse_probe(cen, q, nprobe)
    out = []
    for qi in range(4):
        T = orc.pq_lut(cb, q[qi:qi + 1])[0][:M]                     # [M, 256] fp32
        mn = T.min(1); rng_ = (T.max(1) - mn)
        scale = float(rng_.max() / 255.0)                            # one scale per query (the engine's rule: largest range)
        u8 = np.clip(np.rint((T - mn[:, None]) / scale), 0, 255)
        err = np.abs(T - (mn[:, None] + scale * u8)).max(1)
        eps = float(err.sum())
        s_all, a_all, first = [], [], None
        for j, l in enumerate(pid[qi]):
            codes, ids = ix.get_list(int(l))
            idx = np.arange(M)[None, :]
            s = pdis[qi, j] + T[idx, codes].sum(1)
            a = pdis[qi, j] + (mn.sum() + scale * u8[idx, codes].sum(1))
            s_all.append(s); a_all.append(a)
            if j == 0:
                first = a
        s_all, a_all = np.concatenate(s_all), np.concatenate(a_all)
        srt = np.sort(s_all)[::-1]
        r = {"q": qi, "scanned": int(len(s_all)), "sigma_s": float(s_all.std()), "eps_bound": eps, "max_abs_err": float(np.abs(a_all - s_all).max()),
             "rms_err": float(np.sqrt(((a_all - s_all) ** 2).mean())), "closest_list_len": int(len(first))}
        fs = np.sort(first)[::-1]
        for k in (10, 100, 1000, 2000):
            sk = srt[k - 1]
            r[f"k{k}"] = {"cand_sk_minus_1eps": int((a_all >= sk - eps).sum()), "cand_sk_minus_2eps": int((a_all >= sk - 2 * eps).sum()),
                          "cand_sample2048_kth_minus_2eps": int((a_all >= np.sort(first[:2048])[::-1][min(k, 2047) - 1] - 2 * eps).sum()) if len(first) >= 64 else None,
                          "cand_wholelist_kth_minus_2eps": int((a_all >= fs[min(k, len(fs)) - 1] - 2 * eps).sum()),
                          "cand_actualerr": int((a_all >= sk - np.abs(a_all - s_all).max()).sum())}
        out.append(r)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
