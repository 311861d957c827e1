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


def measure(which, n=0, steps=5, batch=1024, nlist=4096, nprobe=32, check=4, small_batches=True, metric="ip", k=10, extra_ks=(), params=()):
    """One non-headline config on cuda:0 -> result dict (the same object bench.py embeds under "configs")."""
    import torch, rsx
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    n = n or (10_000_000 if which == "flat" else 20_000_000)
    nq = batch
    mcode = 1 if metric == "l2" else 0
    Q = torch.empty((nq * (steps + 1), D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, Q.shape[0], out=Q)
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    t0 = time.time()
    if which == "flat":
        ix = rsx.IndexFlat(D, rsx.METRIC_L2 if mcode else rsx.METRIC_INNER_PRODUCT)
    else:
        ix = rsx.IndexIVFFlat(None, D, nlist, rsx.METRIC_L2 if mcode else rsx.METRIC_INNER_PRODUCT)
        nt = min(n, 256 * nlist)
        xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, 0, nt, out=xt)
        ix.train(xt); del xt
        ix.nprobe = nprobe
        # pass 1: count list sizes with quantizer.assign, reserve exactly (a re-layout of a 153.6 GB index
        # cannot hold two copies in 288 GB); pass 2: add
        counts = np.zeros(nlist, dtype=np.int64)
        for c0 in range(0, n, buf.shape[0]):
            nb = min(buf.shape[0], n - c0)
            rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
            counts += np.bincount(ix.assign(buf[:nb]), minlength=nlist)
        ix.reserve_lists(counts)
    for c0 in range(0, n, buf.shape[0]):
        nb = min(buf.shape[0], n - c0)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
        ix.add(buf[:nb])
    torch.cuda.synchronize()
    build_s = time.time() - t0
    for name, val in params:
        ix.set_param(name, val)
    lm_cache = {}

    def run_k(k):
        ix.search(Q[:nq], k)
        ix.set_param("profile", 2 if which == "ivfflat" else 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(1, steps + 1):
            Dq, Iq = ix.search(Q[s * nq:(s + 1) * nq], k)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        scan_ms = ix.get_timing("scan") / steps
        res = {"config": f"{which} {n}x{D} batch={nq} k={k}" + (f" nlist={nlist} nprobe={nprobe}" + (" metric=L2" if mcode else "") if which == "ivfflat" else f" metric={'L2' if mcode else 'IP'}"),
               "queries_per_s": round(steps * nq / el, 1), "ms_per_step": round(el / steps * 1e3, 3),
               "scan_ms": round(scan_ms, 3), "select_ms": round(ix.get_timing("select") / steps, 3),
               "finalize_ms": round(ix.get_timing("finalize") / steps, 3), "build_s": round(build_s, 1),
               "storage_dtype": ix.storage_dtype,
               "stage_ms": {x: round(ix.get_timing(x) / steps, 4) for x in ("convert", "coarse", "select_probe", "group", "scan0", "select0", "scan", "select", "finalize", "total")},
               "certificate_fallback_queries_per_step": round(ix.get_timing("fallback_queries") / steps, 3)}
        if which == "ivfflat":
            res["filter_overflow_queries_per_step"] = round(ix.get_timing("ivf_filter_overflow_queries") / steps, 3)
        if which == "ivfflat" and ix.get_timing("cand_keys") > 0:      # filtered scan (profile 2): keys that passed the in-kernel filter
            res["filter_keys_per_query"] = {"mean": round(ix.get_timing("cand_keys") / steps / nq, 1), "max": ix.get_timing("cand_keys_max")}
        if which == "flat":
            fl = 2.0 * nq * n * D
            res["roofline"] = {"bound": "mfma", "kernel": "k_flat_gemm2", "achieved": round(fl / (scan_ms * 1e-3) / 1e12, 1), "peak": 2500.0,
                               "unit": "TFLOP/s", "frac": round(fl / (scan_ms * 1e-3) / 2.5e15, 4),
                               "note": "2*nq*N*d flop / scan stage (k_flat_gemm2 + the first chunk's k_select; HIP events on the library stream)"}
        else:
            rows = ix.get_timing("scanned_vectors") / steps
            uniq = ix.get_timing("scanned_unique_vectors") / steps
            qpl = nq * nprobe / max(1, nlist)      # the library's choice of the scan form (api_search.hip): probing queries per list
            kern = ("k_list_scan3 (128 probing queries per group)" if qpl >= 48 and D in (384, 512, 768, 1024) else
                    "k_list_scan2, 8-wave form (64 per group)" if qpl >= 12 else "k_list_scan2 (16 per group)")
            res["roofline"] = {"bound": "hbm", "kernel": kern, "achieved": round(uniq * D * 2 / (scan_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                               "unit": "GB/s", "frac": round(uniq * D * 2 / (scan_ms * 1e-3) / 8e12, 4),
                               "algorithmic_bytes_per_step": uniq * D * 2, "logical_bytes_per_step": rows * D * 2,
                               "effective_GBs": round(rows * D * 2 / (scan_ms * 1e-3) / 1e9, 1),
                               "note": "achieved = rows of every list probed at least once x d x 2 B (each must cross HBM once per batch) / scan "
                                       "stage; logical = sum over (query, probed list) of len*d*2 B (SURVEY 8d), served from one HBM read per "
                                       "group of 16 / 64 / 128 probing queries"}
        ix.set_param("profile", 0)
        # parity spot check against the oracle (exact arithmetic) on a few queries of the last batch
        if check:
            qs = Q[steps * nq:steps * nq + check].cpu().numpy().astype(np.float32)
            if which == "flat":
                best = None
                for c0 in range(0, n, buf.shape[0]):
                    nb = min(buf.shape[0], n - c0)
                    rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
                    Dc, Ic = orc.flat_search(qs, buf[:nb].cpu().numpy().astype(np.float32), k, mcode)
                    Ic = Ic + c0
                    best = (Dc, Ic) if best is None else orc.merge_topk(np.stack([best[0], Dc]), np.stack([best[1], Ic]), mcode)
                ok = bool(np.array_equal(best[1], Iq[:check].cpu().numpy()) and np.array_equal(best[0], Dq[:check].cpu().numpy()))
            else:
                cen = ix.get_centroids()
                if "lm" not in lm_cache:
                    pid, _ = orc.coarse_probe(cen, qs, nprobe)
                    need = np.unique(pid)
                    off = np.zeros(nlist + 1, np.int64); lens = np.zeros(nlist, np.int64); pay = []; ids = []
                    for l in need:
                        v, i = ix.get_list(int(l)); pay.append(v); ids.append(i); lens[l] = len(i)
                    np.cumsum(lens, out=off[1:])

                    class LM: pass
                    lm = LM(); lm.list_off = off; lm.payload = np.concatenate(pay); lm.ids = np.concatenate(ids)
                    lm_cache["lm"] = lm
                Dr, Ir = orc.ivfflat_search(mcode, cen, lm_cache["lm"], qs, nprobe, k)
                ok = bool(np.array_equal(Ir, Iq[:check].cpu().numpy()) and np.array_equal(Dr, Dq[:check].cpu().numpy()))
            res["oracle_parity_ids_and_scores"] = ok
            res["oracle_checked_queries"] = int(check)
        return res

    res = run_k(k)
    if which == "ivfflat":
        ls = ix.list_sizes()
        res["list_length"] = {"min": int(ls.min()), "p50": int(np.percentile(ls, 50)), "p95": int(np.percentile(ls, 95)), "max": int(ls.max())}
    if which == "flat" and small_batches:
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
    # the reference's own n_docs (ric/conf/default.yaml:84: 1000) on the same index
    for kk in extra_ks:
        r2 = run_k(kk)
        res[f"k{kk}"] = {key: r2[key] for key in ("queries_per_s", "ms_per_step", "scan_ms", "select_ms", "finalize_ms", "stage_ms", "filter_keys_per_query", "filter_overflow_queries_per_step", "certificate_fallback_queries_per_step",
                                                 "roofline", "oracle_parity_ids_and_scores", "oracle_checked_queries") if key in r2}
    del ix, buf, Q
    torch.cuda.synchronize()
    return res


def measure_ivfpq(n=100_000_000, M=16, nlist=8192, nprobe=512, ks=(10, 1000), steps=3, batch=1024, check=4, params=()):
    """The reference's shipped IVF-PQ operating point (ric/conf/ivf_pq.yaml:64-78: n_subquantizers 16, ncentroids 8192,
    probe 512, n_docs 1000) on n synthetic vectors: ms per batch for each k, fallbacks, oracle spot check."""
    import torch, rsx
    from oracle import oracle as orc
    dev = torch.device("cuda", 0)
    nq = batch
    t0 = time.time()
    ix = rsx.IndexIVFPQ(None, D, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
    nt = min(n, 256 * nlist)
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    stride = max(1, n // nt)
    for b in range(0, nt, 4096):
        nb = min(4096, nt - b)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, (b * stride) % max(1, n - nb), nb, out=xt[b:b + nb])
    ix.train(xt); del xt
    ix.nprobe = nprobe
    for name, val in params:
        ix.set_param(name, val)
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, n, buf.shape[0]):
        nb = min(buf.shape[0], n - c0)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb]); ix.add(buf[:nb])
    del buf
    torch.cuda.synchronize()
    build_s = time.time() - t0
    Q = torch.empty((nq * (steps + 1), D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, Q.shape[0], out=Q)
    res = {"config": f"ivfpq {n}x{D} M={M} nlist={nlist} nprobe={nprobe} batch={nq}", "build_s": round(build_s, 1),
           "code_layout": {2: "sliced", 1: "rotated"}.get(int(ix._get("pq_layout")), "granule"), "by_k": {}}
    lm = None
    for k in ks:
        ix.search(Q[:nq], k)
        ix.set_param("profile", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(1, steps + 1):
            Dq, Iq = ix.search(Q[s * nq:(s + 1) * nq], k)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / steps
        r = {"ms_per_step": round(el * 1e3, 3), "queries_per_s": round(nq / el, 1),
             "stage_ms": {x: round(ix.get_timing(x) / steps, 4) for x in ("coarse", "select_probe", "lut8", "group", "scan0", "scan", "select", "finalize", "total")},
             "exact_fallback_queries_per_step": round(ix.get_timing("fallback_queries") / steps, 2),
             "of_them_overflows_per_step": round(ix.get_timing("fallback_overflow_queries") / steps, 2),
             "reranked_from_candidate_row_per_step": round(ix.get_timing("second_chance_queries") / steps, 2)}
        r["outside_stages_ms"] = round(r["ms_per_step"] - r["stage_ms"]["total"], 3)
        ix.set_param("profile", 2)
        ix.search(Q[:nq], k)
        r["filter_survivors_per_query"] = {"mean": round(ix.get_timing("cand_keys") / nq, 1), "max": ix.get_timing("cand_keys_max")}
        ix.set_param("profile", 0)
        if check:
            qs = Q[steps * nq:steps * nq + check].cpu().numpy().astype(np.float32)
            cen, cb = ix.get_centroids(), ix.get_codebooks()
            if lm is None:
                pid, _ = orc.coarse_probe(cen, qs, min(nprobe, nlist))
                need = np.unique(pid[pid >= 0])
                ls = ix.list_sizes()
                lens = np.zeros(nlist, np.int64); lens[need] = ls[need]
                off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])

                class LM: pass
                lm = LM(); lm.list_off = off
                lm.payload = np.empty((int(off[-1]), M), np.uint8); lm.ids = np.empty(int(off[-1]), np.int64)
                for l in need:
                    c, i = ix.get_list(int(l)); lm.payload[off[l]:off[l + 1]] = c; lm.ids[off[l]:off[l + 1]] = i
            Dr, Ir = orc.ivfpq_search(cen, cb, lm, qs, nprobe, k)
            r["oracle_parity_ids_and_scores"] = bool(np.array_equal(Ir, Iq[:check].cpu().numpy()) and np.array_equal(Dr, Dq[:check].cpu().numpy()))
            r["oracle_checked_queries"] = int(check)
        res["by_k"][f"k{k}"] = r
    del ix, Q
    torch.cuda.synchronize()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["flat", "ivfflat", "latency", "ivfpq_ref", "largek"])
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--check", type=int, default=4, help="queries verified against the CPU oracle")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="latency / ivfpq_ref / largek: engine parameter A/B")
    ap.add_argument("--ks", default="", help="ivfpq_ref / largek: comma-separated k values")
    ap.add_argument("--k", type=int, default=10, help="flat / ivfflat: results per query")
    ap.add_argument("--metric", default="ip", choices=["ip", "l2"])
    ap.add_argument("--m", type=int, default=0, help="largek: sub-quantisers of the index (default 96)")
    a = ap.parse_args()
    params = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.param]
    if a.which == "latency":
        return latency(a)
    if a.which == "ivfpq_ref":
        ks = tuple(int(t) for t in a.ks.split(",")) if a.ks else (10, 1000)
        print(json.dumps(measure_ivfpq(a.n or 100_000_000, ks=ks, steps=a.steps, check=a.check, params=params)), flush=True)
        return
    if a.which == "largek":      # the headline index (M = 96, nlist 4096, nprobe 32) at the reference's n_docs
        ks = tuple(int(t) for t in a.ks.split(",")) if a.ks else (10, 100, 1000, 2000)
        print(json.dumps(measure_ivfpq(a.n or 100_000_000, a.m or 96, a.nlist, a.nprobe, ks=ks, steps=a.steps, check=a.check, params=params)), flush=True)
        return
    extra = tuple(int(t) for t in a.ks.split(",")) if a.ks else ()
    print(json.dumps(measure(a.which, a.n, a.steps, a.batch, a.nlist, a.nprobe, a.check, metric=a.metric, k=a.k, extra_ks=extra, params=params)), flush=True)


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
    stages = ("convert", "coarse", "select_probe", "lut8", "group", "scan0", "select0", "scan", "select", "finalize", "total")
    ix.search(q[:1], 10)
    res["stage_ms_single_query"] = {s: round(ix.get_timing(s), 4) for s in stages}
    ix.search(q[:1], 100)
    res["stage_ms_single_query_k100"] = {s: round(ix.get_timing(s), 4) for s in stages}
    ix.search(q[:16], 10)
    res["stage_ms_batch16"] = {s: round(ix.get_timing(s), 4) for s in stages}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
