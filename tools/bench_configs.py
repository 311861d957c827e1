#!/usr/bin/env python3
"""Measurements for the non-headline configs of BASELINE.json (parity-test cases, not the driver's bench line):
  C2  Flat  N x 768 (default 10M), batch 1024, k 10  — MFMA-bound:  TFLOP/s = 2*nq*N*d / t
  C3  IVF-Flat N x 768 (default 20M: 100M fp16 = 153.6 GB needs an exact two-pass reserve),
      nlist 4096, nprobe 32                          — HBM-bound:   GB/s   = scanned rows * 1536 B / t
Each prints one JSON line; ids/scores of a sample of queries are checked against the CPU oracle.
usage: bench_configs.py [flat|ivfflat] [--n N] [--steps K]
"""
import argparse, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
D, NC = 768, 4096
SC, SX, SQ = 1234, 10000, 999


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["flat", "ivfflat", "latency"])
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--check", type=int, default=4, help="queries verified against the CPU oracle")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="latency: engine parameter A/B")
    a = ap.parse_args()
    import torch, rsx
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    if a.which == "latency":
        return latency(a)
    n = a.n or (10_000_000 if a.which == "flat" else 20_000_000)
    nq, k = a.batch, 10
    Q = torch.empty((nq * (a.steps + 1), D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, Q.shape[0], out=Q)
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    t0 = time.time()
    if a.which == "flat":
        ix = rsx.IndexFlatIP(D)
    else:
        ix = rsx.IndexIVFFlat(None, D, a.nlist, rsx.METRIC_INNER_PRODUCT)
        nt = min(n, 256 * a.nlist)
        xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, 0, nt, out=xt)
        ix.train(xt); del xt
        ix.nprobe = a.nprobe
        # pass 1: count list sizes with quantizer.assign, reserve exactly (a re-layout of a 153.6 GB index
        # cannot hold two copies in 288 GB); pass 2: add
        counts = np.zeros(a.nlist, dtype=np.int64)
        for c0 in range(0, n, buf.shape[0]):
            nb = min(buf.shape[0], n - c0)
            rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
            counts += np.bincount(ix.assign(buf[:nb]), minlength=a.nlist)
        ix.reserve_lists(counts)
    for c0 in range(0, n, buf.shape[0]):
        nb = min(buf.shape[0], n - c0)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
        ix.add(buf[:nb])
    torch.cuda.synchronize()
    build_s = time.time() - t0
    ix.search(Q[:nq], k)
    ix.set_param("profile", 2 if a.which == "ivfflat" else 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(1, a.steps + 1):
        Dq, Iq = ix.search(Q[s * nq:(s + 1) * nq], k)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    scan_ms = ix.get_timing("scan") / a.steps
    res = {"config": f"{a.which} {n}x{D} batch={nq} k={k}" + (f" nlist={a.nlist} nprobe={a.nprobe}" if a.which == "ivfflat" else ""),
           "queries_per_s": round(a.steps * nq / el, 1), "ms_per_batch": round(el / a.steps * 1e3, 3),
           "scan_ms": round(scan_ms, 3), "select_ms": round(ix.get_timing("select") / a.steps, 3),
           "finalize_ms": round(ix.get_timing("finalize") / a.steps, 3), "build_s": round(build_s, 1),
           "storage_dtype": ix.storage_dtype}
    if a.which == "flat":
        fl = 2.0 * nq * n * D
        res["roofline"] = {"bound": "mfma", "achieved": round(fl / (scan_ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                           "frac": round(fl / (scan_ms * 1e-3) / 2.5e15, 4),
                           "note": "scan stage = k_flat_gemm + per-chunk k_select launches (HIP events on the library stream)"}
    else:
        rows = ix.get_timing("scanned_vectors") / a.steps
        by = rows * D * 2
        res["roofline"] = {"bound": "hbm", "achieved": round(by / (scan_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                           "frac": round(by / (scan_ms * 1e-3) / 8e12, 4), "algorithmic_bytes_per_batch": by,
                           "note": "algorithmic = sum over (query, probed list) of len*d*2 B; list-major grouping reads a list once per group of <=16 queries"}
    if a.which == "flat":
        # small batches stream the database once through the list-scan kernel (HBM-bound): 16 queries per pass
        small = {}
        for b in (1, 16):
            ix.search(Q[:b], k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for s in range(5):
                ix.search(Q[s * b:(s + 1) * b], k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            small[f"batch{b}"] = {"ms": round(ms, 3), "db_GBps": round(n * D * 2 / (ms * 1e-3) / 1e9, 1)}
        res["small_batch"] = small
    # parity spot check against the oracle (exact arithmetic) on a few queries of the last batch
    if a.check:
        qs = Q[a.steps * nq:a.steps * nq + a.check].cpu().numpy().astype(np.float32)
        if a.which == "flat":
            best = None
            for c0 in range(0, n, buf.shape[0]):
                nb = min(buf.shape[0], n - c0)
                rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
                Dc, Ic = orc.flat_search(qs, buf[:nb].cpu().numpy().astype(np.float32), k, 0)
                Ic = Ic + c0
                best = (Dc, Ic) if best is None else orc.merge_topk(np.stack([best[0], Dc]), np.stack([best[1], Ic]))
            ok = bool(np.array_equal(best[1], Iq[:a.check].cpu().numpy()) and np.array_equal(best[0], Dq[:a.check].cpu().numpy()))
        else:
            cen = ix.get_centroids()
            pid, _ = orc.coarse_probe(cen, qs, a.nprobe)
            need = np.unique(pid)
            off = np.zeros(a.nlist + 1, np.int64); lens = np.zeros(a.nlist, np.int64); pay = []; ids = []
            for l in need:
                v, i = ix.get_list(int(l)); pay.append(v); ids.append(i); lens[l] = len(i)
            np.cumsum(lens, out=off[1:])

            class LM: pass
            lm = LM(); lm.list_off = off; lm.payload = np.concatenate(pay); lm.ids = np.concatenate(ids)
            Dr, Ir = orc.ivfflat_search(0, cen, lm, qs, a.nprobe, k)
            ok = bool(np.array_equal(Ir, Iq[:a.check].cpu().numpy()) and np.array_equal(Dr, Dq[:a.check].cpu().numpy()))
        res["oracle_parity_ids_and_scores"] = ok
    print(json.dumps(res), flush=True)


def latency(a):
    """The reference's own protocol (api/api_index.py:88-95): 30 single-query searches, first 10 warm-up, mean of
    the last 20 — here on a 100M x 768 IVF-PQ index (M=96, nlist=4096, nprobe=32, k=10), host-resident query."""
    import torch, rsx
    dev = torch.device("cuda", 0)
    n = a.n or 100_000_000
    ix = rsx.IndexIVFPQ(None, D, a.nlist, 96, 8, rsx.METRIC_INNER_PRODUCT)
    nt = min(n, 256 * a.nlist)
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    rsx.synth_vectors(D, NC, SC, SX, 0.5, 0, nt, out=xt)
    ix.train(xt); del xt
    ix.nprobe = a.nprobe
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, n, buf.shape[0]):
        nb = min(buf.shape[0], n - c0)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
        ix.add(buf[:nb])
    q = rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, 30)   # numpy fp16 on the host
    res = {"config": f"single-query latency, {n}x{D} IVF-PQ M=96 nlist={a.nlist} nprobe={a.nprobe} k=10, host query/result"}

    def run(tag):
        for name, k in (("k10", 10), ("k100", 100)):
            ts = []
            for i in range(30):
                t0 = time.perf_counter(); ix.search(q[i:i + 1], k); ts.append(time.perf_counter() - t0)
            res[f"mean_ms_{name}{tag}"] = round(float(np.mean(ts[10:])) * 1e3, 4)
            res[f"min_ms_{name}{tag}"] = round(float(np.min(ts[10:])) * 1e3, 4)
        for b in (16, 64):
            ts = []
            for i in range(30):
                t0 = time.perf_counter(); ix.search(q[:b] if b <= 30 else np.concatenate([q, q, q])[:b], 10); ts.append(time.perf_counter() - t0)
            res[f"mean_ms_batch{b}_k10{tag}"] = round(float(np.mean(ts[10:])) * 1e3, 4)

    run("")
    for kv in a.param:            # A/B of engine parameters, e.g. --param pq_filter=0
        name, val = kv.split("=")
        ix.set_param(name, int(val))
    if a.param:
        run("_" + ",".join(a.param))
    ix.set_param("profile", 1)
    ix.search(q[:1], 10)
    res["stage_ms_single_query"] = {s: round(ix.get_timing(s), 4) for s in
                                    ("convert", "coarse", "select_probe", "lut8", "group", "scan0", "select0", "scan", "select", "finalize", "total")}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
