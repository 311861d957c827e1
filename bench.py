#!/usr/bin/env python3
"""bench.py — queries/s (+ recall@10) of the MI355X IVF-PQ search path.

Workload (BASELINE.json metric / configs[3]): 100M x 768 IVF-PQ, M = 96, nbits = 8, nlist = 4096,
nprobe = 32, batch = 1024 queries, k = 10, synthetic Gaussian-mixture fp16 embeddings.
One "step" = one rsx_search call over one batch of 1024 queries that is already resident in HBM,
results left in HBM.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: STRONG scaling of the same 100M index — rank r holds the contiguous id range r of N (same
centroids / codebooks everywhere), every rank searches the full batch on its shard, one RCCL
all-gather of the packed [nq, k] candidates, merge kernel on every rank (sharded.ShardedSearcher).

Rank 0 prints ONE JSON line.  The index build (synthesis, training, add) is setup and is not timed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd"))
sys.path.insert(0, REPO)

D = 768
SEED_C, SEED_X, SEED_Q = 1234, 10000, 999
SIGMA, SIGMA_Q = 0.5, 0.1
NCENTRES = 4096
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=100_000_000, help="total vectors in the index (all GPUs)")
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--m", type=int, default=96)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=1_000_000, help="vectors synthesised per add call")
    ap.add_argument("--cpu-queries", type=int, default=128, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-recall", action="store_true", help="skip the exact ground truth (recall = null)")
    ap.add_argument("--diag", action="store_true", help="print a fast-vs-exact comparison of the first timed batch and exit")
    ap.add_argument("--shard", choices=["vectors", "lists"], default="vectors",
                    help="multi-GPU partition of the index: by contiguous id ranges (the reference's shards; default) or by "
                         "inverted lists (rank r owns lists l %% N == r) — experimental: measured slower until the ranks "
                         "exchange their pre-pass thresholds, see DESIGN.md section 6")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="engine parameter for an experiment (rsx_set_param), e.g. pq_filter=0; not for the reported line")
    ap.add_argument("--ab", action="store_true", help="also time the per-pair v1 scan kernel (same process, same index)")
    args = ap.parse_args()

    import torch
    import rsx
    from sharded import ShardedSearcher, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU search path)"
    torch.cuda.set_device(local_rank)
    os.environ["RSX_DEVICE"] = str(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    n_total, nq, k = args.n, args.batch, args.k
    lo, hi = shard_range(n_total, rank, world)
    n_local = hi - lo
    t_setup = time.time()

    # ---------------- train (same deterministic training set on every rank -> identical parameters)
    index = rsx.IndexIVFPQ(rsx.IndexFlatIP(D), D, args.nlist, args.m, 8, rsx.METRIC_INNER_PRODUCT, device=local_rank)
    n_train = min(n_total, 256 * args.nlist)
    xt = torch.empty((n_train, D), dtype=torch.float16, device=dev)
    # training sample = a strided pass over the whole id range (every chunk contributes)
    stride = max(1, n_total // n_train)
    if stride == 1:
        rsx.synth_vectors(D, NCENTRES, SEED_C, SEED_X, SIGMA, 0, n_train, out=xt)
    else:
        blk = 4096
        for b in range(0, n_train, blk):
            nb = min(blk, n_train - b)
            rsx.synth_vectors(D, NCENTRES, SEED_C, SEED_X, SIGMA, (b * stride) % max(1, n_total - nb), nb, out=xt[b:b + nb])
    t0 = time.time()
    index.train(xt)
    del xt
    if rank == 0:
        log(f"train: {time.time() - t0:.1f}s on {n_train} vectors")
    if world > 1:  # belt and braces: every shard must quantise with rank 0's parameters
        cen = torch.from_numpy(index.get_centroids()).to(dev)
        cb = torch.from_numpy(index.get_codebooks()).to(dev)
        dist.broadcast(cen, 0); dist.broadcast(cb, 0)
        fresh = rsx.IndexIVFPQ(rsx.IndexFlatIP(D), D, args.nlist, args.m, 8, rsx.METRIC_INNER_PRODUCT, device=local_rank)
        fresh.set_centroids(cen.cpu().numpy()); fresh.set_codebooks(cb.cpu().numpy())
        index = fresh
    index.nprobe = args.nprobe
    list_shards = world > 1 and args.shard == "lists"
    if list_shards:   # this rank keeps the lists l % world == rank of the whole add stream (same ids as a single index)
        index.set_param("add_list_mod", world)
        index.set_param("add_list_rem", rank)
    for kv in args.param:
        name, val = kv.split("=")
        index.set_param(name, int(val))

    # ---------------- queries (all steps resident in HBM)
    nsteps = args.warmup + args.steps
    Q = torch.empty((nsteps * nq, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NCENTRES, SEED_C, SEED_X, SIGMA, n_total, SEED_Q, SIGMA_Q, 0, nsteps * nq, out=Q)
    Qgt = Q[args.warmup * nq:(args.warmup + 1) * nq]  # recall is measured on the first timed batch

    # ---------------- add (+ streaming exact ground truth with the Flat engine)
    t0 = time.time()
    buf = torch.empty((args.chunk, D), dtype=torch.float16, device=dev)
    flat = None if args.no_recall else rsx.IndexFlatIP(D, device=local_rank)
    gtD = gtI = None
    if list_shards:   # every rank streams the whole id range through add (assignment + its own lists' codes)
        for c0 in range(0, n_total, args.chunk):
            nb = min(args.chunk, n_total - c0)
            rsx.synth_vectors(D, NCENTRES, SEED_C, SEED_X, SIGMA, c0, nb, out=buf[:nb])
            index.add(buf[:nb])
    for c0 in range(lo, hi, args.chunk):
        nb = min(args.chunk, hi - c0)
        if not list_shards or flat is not None:
            rsx.synth_vectors(D, NCENTRES, SEED_C, SEED_X, SIGMA, c0, nb, out=buf[:nb])
        if not list_shards:
            index.add(buf[:nb])
        if flat is not None:
            flat.reset()
            flat.add(buf[:nb])
            Dc, Ic = flat.search(Qgt, k)
            Ic = Ic + c0
            if gtD is None:
                gtD, gtI = Dc, Ic
            else:
                gtD, gtI = rsx.merge_topk(torch.stack([gtD, Dc]), torch.stack([gtI, Ic]))
    del buf, flat
    torch.cuda.synchronize()
    if list_shards:
        n_local = index.ntotal      # the vectors of this rank's lists
    if rank == 0:
        log(f"add: {time.time() - t0:.1f}s for {n_local} vectors/rank; setup total {time.time() - t_setup:.1f}s")
    if any(kv.startswith("add_list_mod=") for kv in args.param):
        n_local = index.ntotal      # experiment: one list shard of an N-way index measured on a single GPU
    assert index.ntotal == n_local

    searcher = ShardedSearcher(index, id_offset=0 if list_shards else lo) if world > 1 else None

    def step(i):
        q = Q[i * nq:(i + 1) * nq]
        if searcher is not None:
            return searcher.search(q, k)
        return index.search(q, k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.diag:
        q = Q[args.warmup * nq:(args.warmup + 1) * nq]
        index.set_param("scan_kernel", 2)
        De, Ie = index.search(q, k)
        index.set_param("scan_kernel", 0)
        index.set_param("profile", 1)
        Df, If = index.search(q, k)
        nfb = index.get_timing("fallback_queries")
        De, Ie, Df, If = De.cpu().numpy(), Ie.cpu().numpy(), Df.cpu().numpy(), If.cpu().numpy()
        badq = np.nonzero((Ie != If).any(1) | (De != Df).any(1))[0]
        log(f"diag: fallback queries {nfb}; queries differing fast vs exact: {len(badq)} of {nq}: {badq[:20].tolist()}")
        for qi in badq[:6]:
            log(f" q={qi}\n  exact I {Ie[qi].tolist()}\n  fast  I {If[qi].tolist()}\n  exact D {De[qi].tolist()}\n  fast  D {Df[qi].tolist()}")
        for kp in (128, 512):
            index.set_param("pq_fast_kp", kp); index.set_param("profile", 1)
            Dg, Ig = index.search(q, k)
            Dg, Ig = Dg.cpu().numpy(), Ig.cpu().numpy()
            nb = int(((Ie != Ig).any(1) | (De != Dg).any(1)).sum())
            log(f"diag: K'={kp}: fallbacks {index.get_timing('fallback_queries')}, differing queries {nb}")
        index.set_param("pq_fast_kp", 0)
        # each query alone through the fast path
        nb1 = 0
        for qi in badq[:4]:
            D1, I1 = index.search(q[qi:qi + 1], k)
            same = np.array_equal(I1.cpu().numpy()[0], Ie[qi]) and np.array_equal(D1.cpu().numpy()[0], De[qi])
            log(f"diag: q={qi} alone through the fast path: equal to exact = {same}")
        return

    for i in range(args.warmup):
        step(i)
    index.set_param("profile", 1)      # HIP events on the library's stream around each stage
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, nsteps):
        out = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms = index.get_timing("scan")
    scan_launches = index.get_timing("scan_launches")
    stage_ms = {s: round(index.get_timing(s) / args.steps, 4) for s in
                ("convert", "coarse", "select_probe", "lut", "lut8", "group", "scan0", "select0", "scan", "select", "finalize", "total")}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fallbacks = index.get_timing("fallback_queries") / max(1.0, index.get_timing("fast_queries"))
    ab = None
    if args.ab:
        ab = {}
        for name, sk in (("exact_list_major_scan2", 2), ("exact_per_pair_scan", 1)):
            index.set_param("scan_kernel", sk)
            for i in range(args.warmup):
                step(i)
            index.set_param("profile", 1)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.warmup, nsteps):
                step(i)
            barrier()
            ab[name] = {"ms_per_step": round((time.perf_counter() - t1) / args.steps * 1e3, 4),
                        "scan_ms_per_step": round(index.get_timing("scan") / args.steps, 4)}
        index.set_param("scan_kernel", 0)

    # algorithmic bytes of the dominant kernel (k_pq_scan2): every (query, probed list) pair reads the
    # list's codes once = M bytes per scanned vector (SURVEY §8d / DESIGN.md).
    index.set_param("profile", 2)
    step(args.warmup)
    scanned = index.get_timing("scanned_vectors")
    cand_keys = index.get_timing("cand_keys")          # keys that passed the in-kernel filter (this one step)
    cand_keys_max = index.get_timing("cand_keys_max")
    dbg_vals = {k: index.get_timing("dbg_" + k) for k in ("hit_blocks", "hit_clk", "loop_clk", "blocks")}
    index.set_param("profile", 0)
    launches_per_step = max(1.0, scan_launches / args.steps)
    scan_bytes = scanned * args.m / launches_per_step          # algorithmic bytes of ONE launch
    ms_per_launch = scan_ms / max(1.0, scan_launches)
    achieved = scan_bytes / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0

    # recall@k of the first timed batch against the exact streaming ground truth (all shards merged)
    D1, I1 = step(args.warmup)
    recall = None
    if gtD is not None:
        if world > 1:
            gD = torch.empty((world,) + tuple(gtD.shape), dtype=gtD.dtype, device=dev)
            gI = torch.empty((world,) + tuple(gtI.shape), dtype=gtI.dtype, device=dev)
            dist.all_gather_into_tensor(gD, gtD.contiguous()); dist.all_gather_into_tensor(gI, gtI.contiguous())
            gtD, gtI = rsx.merge_topk(gD, gI)
        a, b = I1.cpu().numpy(), gtI.cpu().numpy()
        recall = float(np.mean([len(set(x.tolist()) & set(y.tolist())) / k for x, y in zip(a, b)]))

    # ---------------- CPU baseline (rank 0, N = 1): the oracle's FAISS-structured IVFPQ search on the
    # host cores over a bounded sample of the same queries and the same index.
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        from oracle import oracle as orc
        ns = min(args.cpu_queries, nq)
        qs = Qgt[:ns].cpu().numpy().astype(np.float32)
        cen, cb = index.get_centroids(), index.get_codebooks()
        pid, _ = orc.coarse_probe(cen, qs, min(args.nprobe, args.nlist))
        need = np.unique(pid[pid >= 0])
        off = np.zeros(args.nlist + 1, dtype=np.int64)
        payload, ids = [], []
        lens = np.zeros(args.nlist, dtype=np.int64)
        for l in need:
            c, i = index.get_list(int(l))
            payload.append(c); ids.append(i); lens[l] = len(i)
        np.cumsum(lens, out=off[1:])

        class LM:
            pass
        lm = LM()
        lm.list_off = off
        lm.payload = np.concatenate(payload) if payload else np.zeros((0, args.m), np.uint8)
        lm.ids = np.concatenate(ids) if ids else np.zeros(0, np.int64)
        orc.ivfpq_search(cen, cb, lm, qs[:2], args.nprobe, k, heap=True)  # page in
        t0 = time.perf_counter()
        Dc, Ic = orc.ivfpq_search(cen, cb, lm, qs, args.nprobe, k, heap=True)
        cpu_s = time.perf_counter() - t0
        cpu = {"value": round(ns / cpu_s, 3), "unit": "queries/s", "cores": orc.num_threads(), "kind": "port",
               "sample": f"{ns} of the {nq} queries of the first timed batch, same {n_total}-vector index "
                         f"(probed lists copied to host), oracle orc_ivfpq_search_heap, OpenMP over queries"}
        parity = bool(np.array_equal(Ic, I1[:ns].cpu().numpy()) and np.array_equal(Dc, D1[:ns].cpu().numpy()))
        log(f"cpu baseline: {ns} queries in {cpu_s:.2f}s on {orc.num_threads()} threads; parity with GPU ids+scores: {parity}")

    if rank == 0:
        traffic = None
        tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath))
                if t.get("n") == n_total and t.get("n_gpus") == world:
                    traffic = t.get("k_pq_scan8_hbm_bytes_per_launch")
            except Exception:
                pass
        res = {
            "metric": "queries/sec + recall@10, 100M x 768 IVF-PQ nprobe=32 batch=1024",
            "value": round(args.steps * nq / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8 codes; u8-table integer scan, exact f32-table re-rank (certified)",
            "data": "synthetic",
            "recall_at_10": recall,
            "config": {"workload": f"{n_total}x{D} IVF-PQ M={args.m} nbits=8 nlist={args.nlist} nprobe={args.nprobe} "
                                   f"batch={nq} k={k}, inner product, by_residual",
                       "vectors_per_gpu": n_local, "parallelism": (f"index sharded by inverted lists (l % {world} == rank) over {world} GPU(s)" if list_shards
                                       else f"index sharded by id range over {world} GPU(s)")},
            "roofline": {"bound": "hbm", "kernel": "k_pq_scan8", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": scan_bytes, "ms_per_launch": round(ms_per_launch, 4),
                         "launches_per_step": launches_per_step,
                         "traffic_gbs": (round(traffic / (ms_per_launch * 1e-3) / 1e9, 2) if traffic else None),
                         "note": "achieved = scanned code bytes (sum over (query, probed list) of len*M) / HIP-event "
                                 "duration of the scan launch on the library stream, rank 0. frac > 1 is possible: the "
                                 "list-major kernel reads each code byte from HBM once for up to 24 grouped queries, so "
                                 "the algorithmic bytes exceed the measured HBM traffic (`traffic`, PMC FETCH_SIZE); the "
                                 "kernel is bound by LDS table gathers, not HBM (DESIGN.md 4.1)"},
            "stage_ms_per_step": stage_ms,
            "certificate_fallback_fraction": fallbacks,
            "filter_survivors_per_query": {"mean": round(cand_keys / max(1, nq), 1), "max": cand_keys_max},
            "dbg": dbg_vals,
            "ab_exact_kernels_same_process": ab,
            "cpu_baseline": cpu,
            "cpu_parity_ids_and_scores_bit_exact": parity,
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
