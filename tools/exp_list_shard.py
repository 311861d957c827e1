#!/usr/bin/env python3
"""One rank of an N-way LIST-sharded 100M index, measured on one GPU with and without the threshold exchange.
The full index is built beside the shard on the same GPU; "exchanged" thresholds = max(shard's, full index's) — what the
all-reduce(MAX) over the N ranks delivers (the full index's pre-pass sees every rank's closest lists)."""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--ways", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch, rsx
    from sharded import raise_thresholds
    D, NC, nlist, M, nq, k = 768, 4096, 4096, 96, 1024, 10
    dev = torch.device("cuda", 0)
    full = rsx.IndexIVFPQ(None, D, nlist, M, 8, 0)
    nt = 256 * nlist
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    stride = max(1, a.n // nt)
    for b in range(0, nt, 4096):
        rsx.synth_vectors(D, NC, 1234, 10000, 0.5, (b * stride) % (a.n - 4096), 4096, out=xt[b:b + 4096])
    full.train(xt); del xt
    shard = rsx.IndexIVFPQ(None, D, nlist, M, 8, 0)
    shard.set_centroids(full.get_centroids()); shard.set_codebooks(full.get_codebooks())
    shard.set_param("add_list_mod", a.ways); shard.set_param("add_list_rem", 0)
    full.nprobe = shard.nprobe = 32
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, a.n, 1_000_000):
        rsx.synth_vectors(D, NC, 1234, 10000, 0.5, c0, 1_000_000, out=buf); full.add(buf); shard.add(buf)
    del buf
    Q = torch.empty(((a.steps + 2) * nq, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, 1234, 10000, 0.5, a.n, 999, 0.1, 0, Q.shape[0], out=Q)
    out = {"shard_vectors": shard.ntotal, "ways": a.ways}

    wall = [0.0]

    def run(exchange):
        for i in range(2):
            go(i, exchange)
        shard.set_param("profile", 2)
        wall[0] = 0.0
        for i in range(2, 2 + a.steps):
            go(i, exchange)
        el = wall[0] / a.steps
        r = {"ms_per_batch_wall": round(el * 1e3, 3),
             "scan_ms": round(shard.get_timing("scan") / a.steps, 3), "select_ms": round(shard.get_timing("select") / a.steps, 3), "exact_fallbacks_per_batch": shard.get_timing("fallback_queries") / a.steps,
             "reranked_per_batch": shard.get_timing("second_chance_queries") / a.steps,
             "candidates_per_query": round(shard.get_timing("cand_keys") / a.steps / nq, 1)}
        shard.set_param("profile", 0)
        return r

    def go(i, exchange):
        q = Q[i * nq:(i + 1) * nq]
        torch.cuda.synchronize()
        if not exchange:
            t0 = time.perf_counter(); r = shard.search(q, k); torch.cuda.synchronize(); wall[0] += time.perf_counter() - t0
            return r
        tf = full.search_prepass(q, k)          # stands in for the other ranks (untimed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = shard.search_prepass(q, k)
        raise_thresholds([tf, ts])              # the all-reduce(MAX): here a device-side maximum of two vectors
        torch.cuda.synchronize()
        r = shard.search_scan()
        torch.cuda.synchronize(); wall[0] += time.perf_counter() - t0
        full.search_scan()
        return r
    out["own_thresholds"] = run(False)
    out["exchanged_thresholds"] = run(True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
