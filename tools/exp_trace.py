#!/usr/bin/env python3
"""Sibling start-skew measurement of k_pq_scan_rot (needs the -DRSX_MEASURE build: RSX_LIB=.../librsx_measure.so).
Builds an index, runs a few batches, pulls the per-item trace of the LAST scan launch (start / end ticks of 10 ns, workgroup,
family window) and prints: item duration quantiles, start skew between consecutive members of a family, fraction of
followers that start within T us of the family's first starter, brake polls used."""
import argparse, ctypes, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--pace", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    import torch, rsx
    D, NC = 768, 4096
    dev = torch.device("cuda", 0)
    ix = rsx.IndexIVFPQ(rsx.IndexFlatIP(D), D, 4096, 96, 8, rsx.METRIC_INNER_PRODUCT, device=0)
    nt = min(args.n, 256 * 4096)
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    stride = max(1, args.n // nt)
    for b in range(0, nt, 4096):
        nb = min(4096, nt - b)
        rsx.synth_vectors(D, NC, 1234, 10000, 0.5, (b * stride) % max(1, args.n - nb), nb, out=xt[b:b + nb])
    ix.train(xt); del xt
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, args.n, 1_000_000):
        nb = min(1_000_000, args.n - c0)
        rsx.synth_vectors(D, NC, 1234, 10000, 0.5, c0, nb, out=buf[:nb]); ix.add(buf[:nb])
    del buf
    ix.nprobe = 32
    ix.set_param("pq_pace", args.pace)
    Q = torch.empty((4 * args.batch, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, 1234, 10000, 0.5, args.n, 999, 0.1, 0, 4 * args.batch, out=Q)
    for i in range(4):
        ix.search(Q[i * args.batch:(i + 1) * args.batch], 10)
    torch.cuda.synchronize()
    NI = 65536
    tr = np.zeros((NI, 4), dtype=np.uint64)
    L = rsx.lib()
    assert L.rsx_debug_rot_trace(tr.ctypes.data_as(ctypes.c_void_p), NI) == 0
    start = (tr[:, 0]).astype(np.int64); end = (tr[:, 1] & np.uint64((1 << 56) - 1)).astype(np.int64)
    polls = (tr[:, 1] >> np.uint64(56)).astype(np.int64)
    blk = (tr[:, 2] >> np.uint64(32)).astype(np.int64); nit = (tr[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    f0 = (tr[:, 3] >> np.uint64(32)).astype(np.int64); f1 = (tr[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    valid = (start > 0) & (end >= start)
    n = int(valid.nonzero()[0].max()) + 1 if valid.any() else 0
    t0 = start[valid].min()
    dur = (end - start)[valid] * 0.01
    out = {"items": int(valid.sum()), "item_us": {q: round(float(np.percentile(dur, q)), 1) for q in (5, 25, 50, 75, 95)},
           "scan_span_us": round(float((end[valid].max() - t0) * 0.01), 1)}
    # family analysis (families as the kernel saw them: [f0, f1) windows; take items whose window starts at themselves as leaders)
    skew_next, lag_first, fam_sizes = [], [], []
    i = 0
    while i < n:
        if not valid[i]:
            i += 1; continue
        a, b = int(f0[i]), int(f1[i])
        if b - a <= 1 or a != i:
            i += 1; continue
        mem = [j for j in range(a, min(b, n)) if valid[j]]
        st = np.sort(start[mem])
        fam_sizes.append(len(mem))
        for j in range(1, len(st)):
            skew_next.append((st[j] - st[j - 1]) * 0.01)
            lag_first.append((st[j] - st[0]) * 0.01)
        i = b
    skew_next, lag_first = np.array(skew_next), np.array(lag_first)
    out["families"] = len(fam_sizes); out["mean_family"] = round(float(np.mean(fam_sizes)), 2) if fam_sizes else 0
    if len(lag_first):
        out["follower_lag_behind_first_us"] = {q: round(float(np.percentile(lag_first, q)), 1) for q in (10, 25, 50, 75, 90)}
        out["followers_within_us"] = {t: round(float((lag_first <= t).mean()), 3) for t in (1, 2, 3, 5, 8, 12, 20, 40)}
        out["gap_between_consecutive_starts_us"] = {q: round(float(np.percentile(skew_next, q)), 1) for q in (10, 50, 90)}
    # per-workgroup timeline of the LAST launch: busy time (start..end of items as wave 0 sees them) vs gaps between items
    last = valid & (start >= start[valid].max() - 400000)           # within 4 ms of the latest start
    ks = np.nonzero(last)[0]
    lt0, lt1 = start[ks].min(), end[ks].max()
    out["last_launch"] = {"items": int(len(ks)), "span_us": round(float((lt1 - lt0) * 0.01), 1)}
    busy, gaps, nitems, first, lastend = [], [], [], [], []
    for b in np.unique(blk[ks]):
        it = ks[blk[ks] == b]
        it = it[np.argsort(start[it])]
        busy.append(float((end[it] - start[it]).sum() * 0.01))
        gaps.append(float((start[it][1:] - end[it][:-1]).sum() * 0.01))
        nitems.append(len(it)); first.append(float((start[it][0] - lt0) * 0.01)); lastend.append(float((lt1 - end[it][-1]) * 0.01))
    out["last_launch"].update({"workgroups": len(busy), "busy_us_mean": round(float(np.mean(busy)), 1), "gap_us_mean": round(float(np.mean(gaps)), 1),
                               "items_per_wg_mean": round(float(np.mean(nitems)), 1), "first_start_us_mean": round(float(np.mean(first)), 1),
                               "idle_tail_us_mean": round(float(np.mean(lastend)), 1), "idle_tail_us_max": round(float(np.max(lastend)), 1),
                               "gap_per_item_us": round(float(np.sum(gaps) / max(1, np.sum(nitems) - len(nitems))), 2)})
    xc = blk[ks] & 7
    out["per_xcd"] = {"items": [int((xc == x).sum()) for x in range(8)],
                      "busy_ms": [round(float(((end[ks] - start[ks])[xc == x]).sum() * 1e-5), 2) for x in range(8)],
                      "finish_us": [round(float((end[ks][xc == x].max() - lt0) * 0.01), 1) for x in range(8)],
                      "first_wg_done_us": [round(float(min(end[ks[(blk[ks] == b)]].max() for b in np.unique(blk[ks][xc == x])) - lt0) * 0.01, 1) for x in range(8)]}
    wv = np.zeros((16384, 16), dtype=np.uint32)
    if hasattr(L, "rsx_debug_rot_wave") and L.rsx_debug_rot_wave(wv.ctypes.data_as(ctypes.c_void_p), 16384) == 0:
        sel = ks[(ks < 16384)]
        sel = sel[nit[sel] >= 8]                  # full-size items only
        w_us = wv[sel].astype(np.float64) * 0.01
        out["wave_scan_us_mean_by_wave"] = [round(float(x), 1) for x in w_us.mean(0)]
        out["wave_scan_us_min_max_mean"] = [round(float(w_us.min(1).mean()), 1), round(float(w_us.max(1).mean()), 1)]
        order = np.argsort(w_us, axis=1)
        out["mean_finish_rank_by_wave"] = [round(float(np.argsort(order, axis=1)[:, w].mean()), 1) for w in range(16)]
    out["brake_polls_per_item_mean"] = round(float(polls[valid].mean()), 2)
    out["iterations_per_item_median"] = int(np.median(nit[valid]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
