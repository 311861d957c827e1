#!/usr/bin/env python3
"""bench.py — queries/s (+ recall@10) of the MI355X IVF-PQ search path.

Workload (BASELINE.json metric / configs[3]): 100M x 768 IVF-PQ, M = 96, nbits = 8, nlist = 4096,
nprobe = 32, batch = 1024 queries, k = 10, synthetic Gaussian-mixture fp16 embeddings.
One "step" = one rsx_search call over one batch of 1024 queries that is already resident in HBM,
results left in HBM (`value`); the same loop with host-resident queries/results is reported beside it as
`pcie_inclusive`, never as `value`.

At N = 1 the line also carries (rank 0, after the timed region, each on its own index): `configs` = BASELINE configs 2
and 3 (Flat 10M batch 1024; IVF-Flat 100M nlist 4096 nprobe 32) with ms/step, roofline and an oracle spot check,
`recall_informative` = recall@10 on a second mixture whose neighbours PQ can resolve, the list-length histogram and
the CPU baseline (the oracle on all host cores, all 1024 queries, three repeats).  --no-configs / --cpu-queries 0 /
--no-recall switch the extras off for experiments.

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
    ap.add_argument("--cpu-queries", type=int, default=1024, help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-recall", action="store_true", help="skip the exact ground truth (recall = null)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 2/3 and the second recall figure")
    ap.add_argument("--no-faiss", action="store_true", help="skip the opportunistic FAISS leg (it only runs when `import faiss` works)")
    ap.add_argument("--diag", action="store_true", help="print a fast-vs-exact comparison of the first timed batch and exit")
    ap.add_argument("--shard", choices=["vectors", "lists"], default="vectors",
                    help="multi-GPU partition of the index: by contiguous id ranges (the reference's shards; default) or by "
                         "inverted lists (rank r owns lists l %% N == r; the ranks all-reduce(MAX) their pre-pass thresholds between the "
                         "two calls of the search, see DESIGN.md section 6)")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="engine parameter for an experiment (rsx_set_param), e.g. pq_filter=0; not for the reported line")
    ap.add_argument("--ab", action="store_true", help="also time the per-pair v1 scan kernel (same process, same index)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI, the default) or gloo (the same collectives routed "
                         "through the host — what lets N ranks share one GPU in the dry-run test; RCCL refuses two ranks per device)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the SAME --n vectors cut over the N GPUs (default: BASELINE's metric is quoted on one 100M index at "
                         "1/2/4/8 GPUs); weak = --n vectors PER GPU (N x --n in all; SURVEY 8(d) C5 asks for both curves)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="dry run: rank r uses device r %% (visible GPUs) instead of requiring one GPU per rank (tests/test_gpu_bench_ranks.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # idle OpenMP workers SLEEP between parallel regions instead of spinning (read when libgomp initialises, i.e. at `import torch`): with the
    # default policy the 128 threads of the CPU-baseline leg kept the host cores busy into the next GPU leg, whose launches then took 14 ms
    # per batch instead of 3.7 (round 5; the CPU leg itself is one ~2 s parallel region per repeat and does not notice)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    import rsx
    from sharded import ShardedSearcher, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU search path)"
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    os.environ["RSX_DEVICE"] = str(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    on_host = world > 1 and args.dist_backend != "nccl"     # gloo: every collective below goes through host memory

    def bcast0(t):                       # rank 0's tensor to everyone, in place
        if on_host:
            h = t.cpu(); dist.broadcast(h, 0); t.copy_(h)
        else:
            dist.broadcast(t, 0)

    def gather_all(t):                   # [world, *t.shape] on every rank
        flat = (t.cpu() if on_host else t).contiguous().view(-1)      # flat views: gloo's all-gather wants [world * n], not [world, ...]
        out = torch.empty(world * flat.numel(), dtype=t.dtype, device=flat.device)
        dist.all_gather_into_tensor(out, flat)
        return out.view((world,) + tuple(t.shape)).to(dev)

    n_total, nq, k = (args.n * world if args.scaling == "weak" else args.n), args.batch, args.k
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
        bcast0(cen); bcast0(cb)
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
    # ... and on a second batch that asks an answerable question of THIS index: queries = a base vector + 0.02 noise (the 0.1
    # noise of the prescribed batch leaves ranks 2..10 to chance at PQ resolution: DESIGN.md 5)
    Qlow = torch.empty((nq, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NCENTRES, SEED_C, SEED_X, SIGMA, n_total, SEED_Q + 1, 0.02, 0, nq, out=Qlow)
    Qgt2 = torch.cat([Qgt, Qlow], 0)

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
            Dc, Ic = flat.search(Qgt2, k)
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

    searcher = ShardedSearcher(index, id_offset=0 if list_shards else lo, exchange_thresholds=list_shards) if world > 1 else None

    def step(i):
        q = Q[i * nq:(i + 1) * nq]
        if searcher is not None:
            return searcher.search(q, k)
        return index.search(q, k)

    def step_q(q):
        return searcher.search(q, k) if searcher is not None else index.search(q, k)

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
        for kp in (128, 512, 1024):
            index.set_param("pq_fast_kp", kp); index.set_param("profile", 1)
            Dg, Ig = index.search(q, k)
            Dg, Ig = Dg.cpu().numpy(), Ig.cpu().numpy()
            bq = np.nonzero((Ie != Ig).any(1) | (De != Dg).any(1))[0]
            nb = len(bq)
            log(f"diag: K'={kp}: fallbacks {index.get_timing('fallback_queries')}, differing queries {nb}: {bq[:8].tolist()}")
            for qi in bq[:3]:
                log(f" q={qi}\n  exact I {Ie[qi].tolist()}\n  K'    I {Ig[qi].tolist()}\n  exact D {De[qi].tolist()}\n  K'    D {Dg[qi].tolist()}")
            if nb:
                index.set_param("pq_fast_kp", kp); index.set_param("profile", 2)
                index.search(q, k)
                log(f"diag: K'={kp}: survivors/query mean {index.get_timing('cand_keys') / nq:.0f} max {index.get_timing('cand_keys_max'):.0f}")
                for qi in bq[:3]:
                    D1_, I1_ = index.search(q[qi:qi + 1], k)
                    log(f"diag: K'={kp} q={qi} alone: equal to exact = {np.array_equal(I1_.cpu().numpy()[0], Ie[qi]) and np.array_equal(D1_.cpu().numpy()[0], De[qi])}")
        index.set_param("pq_fast_kp", 0)
        # each query alone through the fast path
        nb1 = 0
        for qi in badq[:4]:
            D1, I1 = index.search(q[qi:qi + 1], k)
            same = np.array_equal(I1.cpu().numpy()[0], Ie[qi]) and np.array_equal(D1.cpu().numpy()[0], De[qi])
            log(f"diag: q={qi} alone through the fast path: equal to exact = {same}")
        return

    # Settling: the first few dozen batches after the build run ~1.5 % slower than the steady state (measured: 2.656 ms per batch timed behind
    # 5 warm-up steps, 2.623 behind 40 — workspace first touches, TLB and clock settling); SETTLE untimed batches (~0.1 s) precede the W
    # warm-up steps the driver asked for, so that the K timed steps measure the steady state whatever W is.  Reported as "settle_batches".
    SETTLE = 32
    for i in range(SETTLE):
        step(i % nsteps)
    for i in range(args.warmup):
        step(i)
    # The timed region: HIP events on the library's stream around the dominant scan launch only (profile -1) — every event record is ~5 us of
    # stream time, and a mark per stage (profile 1: eight of them) is 1.5 % of a batch.  The per-stage breakdown comes from a second pass
    # over the same batches in profile 1, outside the timed region.
    index.set_param("profile", -1)
    if searcher is not None:
        searcher.profile = True        # ... and CUDA events around pack / collective / merge of the exchange
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, nsteps):
        out = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms = index.get_timing("scan")
    scan_launches = index.get_timing("scan_launches")
    stages_in_timed_region = False
    if not scan_ms > 0.0:              # a path without the scan-only marks (other index kinds, exact kernels): time it with every stage marked
        index.set_param("profile", 1)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.warmup, nsteps):
            out = step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        scan_ms = index.get_timing("scan")
        scan_launches = index.get_timing("scan_launches")
        stages_in_timed_region = True
    fallbacks_timed = index.get_timing("fallback_queries") / max(1.0, index.get_timing("fast_queries"))
    if not stages_in_timed_region:
        index.set_param("profile", 1)
        barrier()
        for i in range(args.warmup, nsteps):
            step(i)
        barrier()
    stage_ms = {s: round(index.get_timing(s) / args.steps, 4) for s in
                ("convert", "coarse", "select_probe", "lut", "lut8", "group", "scan0", "select0", "scan", "select", "finalize", "total")}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fallbacks = fallbacks_timed
    index.set_param("profile", 0)
    if searcher is not None:           # the exchange of this rank (rank 0's are printed): where a multi-GPU step's time goes beside the search
        ex = searcher.stage_ms()
        searcher.profile = False
        for key in ("pack", "collective", "merge"):
            stage_ms["exchange_" + key] = round(ex[key] / max(1, ex["calls"]), 4)

    # ---- the same loop with host-resident queries and results (H2D of Q, D2H of D, I inside the timed region)
    pcie = None
    if world == 1:
        Qh = Q.cpu().numpy()
        for i in range(min(2, args.warmup)):
            index.search(Qh[i * nq:(i + 1) * nq], k)
        t1 = time.perf_counter()
        for i in range(args.warmup, nsteps):
            index.search(Qh[i * nq:(i + 1) * nq], k)
        el_h = time.perf_counter() - t1
        pcie = {"value": round(args.steps * nq / el_h, 2), "unit": "queries/s", "ms_per_step": round(el_h / args.steps * 1e3, 4),
                "note": "numpy fp16 queries in, numpy D/I out: 1.57 MB H2D + 0.12 MB D2H per step inside the timed loop"}
        del Qh

    # ---- the reference's call shape (src/search.py:296: ALL queries in one index.search): four batches in one call (the internal
    # batches run back to back without the round trip through Python).  A side key, never `value`.
    one_call = None
    nb4 = min(4, args.steps)
    if world == 1 and nb4 >= 2 and not args.no_configs:      # (--no-configs: the profiled runs must hold the timed loop's kernels only)
        Q4 = Q[args.warmup * nq:(args.warmup + nb4) * nq]
        index.search(Q4, k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            index.search(Q4, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / 3 * 1e3
        one_call = {"queries": nb4 * nq, "k": k, "ms": round(ms, 4), "queries_per_s": round(nb4 * nq / ms * 1e3, 1),
                    "note": "one index.search call of %d queries (query_batch 1024)" % (nb4 * nq)}
        del Q4

    ab = None
    if args.ab:
        ab = {}
        for name, sk in (("exact_scan_kernel_2", 2), ("exact_scan_kernel_1", 1)):
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

    # ---- work of one scan launch, counted from the actual probe lists of one step (profile 2 reads them back)
    index.set_param("profile", 2)
    step(args.warmup)
    scanned = index.get_timing("scanned_vectors")               # sum over (query, probed list) of len
    scanned_unique = index.get_timing("scanned_unique_vectors")  # vectors of every list probed at least once
    scanned_group = index.get_timing("scanned_group_vectors")    # vectors x groups of <= 4 probing queries
    cand_keys = index.get_timing("cand_keys")                    # keys that passed the in-kernel filter
    cand_keys_max = index.get_timing("cand_keys_max")
    gq = int(index.get_timing("scan_group_queries") or 4)       # queries per table gather of the scan kernel
    index.set_param("profile", 0)
    launches_per_step = max(1.0, scan_launches / args.steps)
    ms_per_launch = scan_ms / max(1.0, scan_launches)
    sec = ms_per_launch * 1e-3
    layout = index._get("pq_layout")                              # 2 = sliced (M = 96 default), 1 = rotated, 0 = granule
    rot = layout in (1, 2)
    kernel = "k_pq_scan_sl8" if layout == 2 else "k_pq_scan_rot" if rot else "k_pq_scan8"

    # HBM roofline: the codes of every list probed at least once must cross HBM once per batch (list-major scan) — that is the
    # kernel's algorithmic HBM traffic; PMC FETCH_SIZE (`traffic`) shows what actually crossed.
    alg_bytes = scanned_unique * args.m / launches_per_step
    achieved = alg_bytes / sec / 1e9 if sec > 0 else 0.0
    # LDS side: one table gather per lane per (vector, sub-quantiser, query group): 8 bytes (ds_read_b64, peak 256 B/clk/CU) for the
    # eight-query scans, 4 bytes (ds_read_b32, peak 128 B/clk/CU) for the four-query ones
    gather_bytes = 8 if gq == 8 else 4
    lds_bytes = scanned_group * args.m * gather_bytes / launches_per_step
    lds_peak = (256.0 if gq == 8 else 128.0) * 256 * 2.4e9 / 1e9

    # recall@k of the first timed batch against the exact streaming ground truth (all shards merged)
    D1, I1 = step(args.warmup)
    import hashlib
    result_sha = hashlib.sha256(D1.cpu().numpy().tobytes() + I1.cpu().numpy().tobytes()).hexdigest()   # equal for every N and both partitions
    recall = None
    recall_low = None
    if gtD is not None:
        if world > 1:
            gtD, gtI = rsx.merge_topk(gather_all(gtD), gather_all(gtI))
        gt = gtI.cpu().numpy()
        a, b = I1.cpu().numpy(), gt[:nq]
        recall = float(np.mean([len(set(x.tolist()) & set(y.tolist())) / k for x, y in zip(a, b)]))
        _, I2 = step_q(Qlow)
        a2, b2 = I2.cpu().numpy(), gt[nq:]
        by_np = {}
        for npb in sorted({1, 8, args.nprobe}):      # the metric's second half read off the benchmarked index itself (task 8)
            index.nprobe = npb
            _, Ix = step_q(Qlow)
            ax = Ix.cpu().numpy()
            by_np[f"nprobe{npb}"] = {"recall_at_10": round(float(np.mean([len(set(x.tolist()) & set(y.tolist())) / k for x, y in zip(ax, b2)])), 4),
                                     "recall_at_1": round(float(np.mean(ax[:, 0] == b2[:, 0])), 4)}
        index.nprobe = args.nprobe
        recall_low = {"recall_at_10": round(float(np.mean([len(set(x.tolist()) & set(y.tolist())) / k for x, y in zip(a2, b2)])), 4),
                      "by_nprobe": by_np,
                      "recall_at_1": round(float(np.mean(a2[:, 0] == b2[:, 0])), 4),
                      "recall_at_1_prescribed_queries": round(float(np.mean(a[:, 0] == b[:, 0])), 4),
                      "queries": "base vector + 0.02 noise (seed 1000), same 100M index, same nprobe; ground truth = exact streaming Flat search"}

    ls = index.list_sizes()
    hist = {"lists": int(len(ls)), "empty": int((ls == 0).sum()), "min": int(ls.min()), "p5": int(np.percentile(ls, 5)),
            "p25": int(np.percentile(ls, 25)), "p50": int(np.percentile(ls, 50)), "p75": int(np.percentile(ls, 75)),
            "p95": int(np.percentile(ls, 95)), "max": int(ls.max()), "mean": round(float(ls.mean()), 1)}

    # ---------------- CPU baseline (rank 0, N = 1): the oracle's FAISS-structured IVFPQ search on all host cores over the
    # queries of the first timed batch and the same index (probed lists copied to the host), three repeats.
    cpu = None
    parity = None
    lm = None
    if rank == 0 and world == 1 and args.cpu_queries > 0:
        from oracle import oracle as orc
        ns = min(args.cpu_queries, nq)
        qs = Qgt[:ns].cpu().numpy().astype(np.float32)
        cen, cb = index.get_centroids(), index.get_codebooks()
        pid, _ = orc.coarse_probe(cen, qs, min(args.nprobe, args.nlist))
        need = np.unique(pid[pid >= 0])
        lens = np.zeros(args.nlist, dtype=np.int64)
        lens[need] = ls[need]
        off = np.zeros(args.nlist + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])

        class LM:
            pass
        lm = LM()
        lm.list_off = off
        lm.payload = np.empty((int(off[-1]), args.m), np.uint8)
        lm.ids = np.empty(int(off[-1]), np.int64)
        for l in need:
            c, i = index.get_list(int(l))
            lm.payload[off[l]:off[l + 1]] = c; lm.ids[off[l]:off[l + 1]] = i
        orc.ivfpq_search(cen, cb, lm, qs[:min(ns, 2 * orc.num_threads())], args.nprobe, k, heap=True)  # page in
        times = []
        for rep in range(3):
            t0 = time.perf_counter()
            Dc, Ic = orc.ivfpq_search(cen, cb, lm, qs, args.nprobe, k, heap=True)
            times.append(time.perf_counter() - t0)
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            model = "unknown"
        cpu = {"value": round(ns / float(np.mean(times)), 3), "unit": "queries/s", "best": round(ns / min(times), 3),
               "repeats_s": [round(t, 3) for t in times], "cores": orc.num_threads(), "cpu_model": model, "kind": "port",
               "sample": f"all {ns} queries of the first timed batch, same {n_total}-vector index ({len(need)} probed lists copied to "
                         f"host), oracle orc_ivfpq_search_heap, OpenMP over queries, mean of 3 repeats"}
        # parity against the oracle's CANONICAL order (score desc, id asc — what the GPU path emits).  The timed variant above is
        # the FAISS-structured heap, whose order among EXACTLY equal scores is heap mechanics (unpinned): queries
        # where it differs from the canonical result by such a tie are counted, not failed.
        Dk, Ik = orc.ivfpq_search(cen, cb, lm, qs, args.nprobe, k)
        Ig, Dg = I1[:ns].cpu().numpy(), D1[:ns].cpu().numpy()
        parity = bool(np.array_equal(Ik, Ig) and np.array_equal(Dk, Dg))
        heap_tie_queries = int(((Ic != Ik).any(1)).sum())
        heap_scores_equal = bool(np.array_equal(Dc, Dk))
        if not parity:
            badq = np.nonzero((Ik != Ig).any(1) | (Dk != Dg).any(1))[0]
            log(f"cpu/gpu differ on {len(badq)} of {ns} queries: {badq[:16].tolist()}")
            for qi in badq[:4]:
                log(f" q={qi}\n  cpu I {Ik[qi].tolist()}\n  gpu I {Ig[qi].tolist()}\n  cpu D {Dk[qi].tolist()}\n  gpu D {Dg[qi].tolist()}")
        cpu["heap_variant_vs_canonical"] = {"queries_with_a_tie_in_different_order": heap_tie_queries, "scores_identical": heap_scores_equal}
        log(f"cpu baseline: {ns} queries x3 in {[round(t, 2) for t in times]} s on {orc.num_threads()} threads; parity with GPU ids+scores: {parity}")

    # ---------------- the reference's own n_docs on the headline index (ric/conf/default.yaml:84, ivf_pq.yaml:78: n_docs 1000;
    # scripts/post_procress.sh:2: 2000): ms per batch, fallbacks, and an oracle spot check
    ops = None
    if rank == 0 and world == 1 and not args.no_configs:
        ops = {}
        for kk in (100, 1000, 2000):
            nrep = min(5, args.steps)
            for i in range(2):                      # untimed: workspaces of this k, and of its (rare) exact re-runs, get allocated here
                index.search(Q[i * nq:(i + 1) * nq], kk)
            index.set_param("profile", 1)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for i in range(args.warmup, args.warmup + nrep):
                Dk_, Ik_ = index.search(Q[i * nq:(i + 1) * nq], kk)
            torch.cuda.synchronize(); el_k = (time.perf_counter() - t1) / nrep
            r = {"ms_per_step": round(el_k * 1e3, 3), "queries_per_s": round(nq / el_k, 1),
                 "exact_fallback_queries_per_step": round(index.get_timing("fallback_queries") / nrep, 2),
                 "reranked_from_candidate_row_per_step": round(index.get_timing("second_chance_queries") / nrep, 2),
                 "stage_ms": {s_: round(index.get_timing(s_) / nrep, 4) for s_ in ("scan0", "scan", "select", "finalize", "total")}}
            index.set_param("profile", 0)
            if cpu is not None:       # the probed lists of the first timed batch are still on the host: 8 queries against the oracle
                Dg_, Ig_ = index.search(Qgt[:8], kk)
                Do_, Io_ = orc.ivfpq_search(cen, cb, lm, qs[:8], args.nprobe, kk)
                r["oracle_parity_ids_and_scores"] = bool(np.array_equal(Io_, Ig_.cpu().numpy()) and np.array_equal(Do_, Dg_.cpu().numpy()))
                r["oracle_checked_queries"] = 8
            ops[f"k{kk}"] = r
            log(f"headline index at k={kk}: {r}")
    lm = None

    # ---------------- the same operating point on other data distributions (tools/bench_dist.py; side keys, never `value`):
    # a hot-list batch on THIS index now, the informative and the norm-skewed mixtures on their own 100M indexes below
    dist_legs = None
    if rank == 0 and world == 1 and not args.no_configs:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_dist
        dist_legs = {}
        try:
            dist_legs["hot_lists"] = bench_dist.hot_list_leg(index, n_total, M=args.m, nlist=args.nlist, nprobe=args.nprobe, k=k,
                                                             steps=min(5, args.steps), batch=nq, check=8 if cpu is not None else 0)
            dist_legs["hot_probe_sets"] = bench_dist.hot_list_leg(index, n_total, M=args.m, nlist=args.nlist, nprobe=args.nprobe, k=k,
                                                                  steps=min(5, args.steps), batch=nq, check=8 if cpu is not None else 0,
                                                                  shared_probe_sets=True)
        except Exception as e:
            dist_legs["hot_lists"] = {"error": repr(e)}
        log(f"hot-list batches: {json.dumps(dist_legs)[:600]}")

    # ---------------- opportunistic FAISS leg: the real reference engine on the same index and queries, when the
    # box has it (tools/faiss_leg.py; never required, {"available": false} otherwise)
    faiss_res = None
    if rank == 0 and world == 1 and not args.no_faiss:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from faiss_leg import faiss_leg
        faiss_res = faiss_leg(index, Qgt.cpu().numpy(), k, args.nprobe, D1.cpu().numpy(), I1.cpu().numpy(), log=log)

    # ---------------- extras on their own indexes (N = 1): BASELINE configs 2 / 3, a second recall figure
    configs = None
    recall2 = None
    if rank == 0 and world == 1 and not args.no_configs:
        index = Q = Qgt = out = D1 = I1 = gtD = gtI = None   # the 100M index leaves HBM before the next ones are built
        import gc
        gc.collect(); torch.cuda.synchronize()
        for leg in ("informative", "norm_skew"):
            t0 = time.time()
            try:
                dist_legs[leg] = bench_dist.mixture_leg(leg, n=n_total, M=args.m, nlist=args.nlist, nprobe=args.nprobe, k=k,
                                                        steps=min(5, args.steps), batch=nq, check=8 if cpu is not None else 0)
            except Exception as e:
                dist_legs[leg] = {"error": repr(e)}
            log(f"distribution leg {leg}: {time.time() - t0:.1f}s -> {json.dumps(dist_legs[leg])[:400]}")
            gc.collect()
        inf = dist_legs.get("informative", {})
        if "recall_at_10" in inf:      # (rounds 1-4 reported this figure from a 10M index of the same mixture)
            recall2 = {"recall_at_10": inf["recall_at_10"], "by_nprobe": {k_: v_["recall_at_10"] for k_, v_ in inf["recall_by_nprobe"].items()},
                       "data": inf["data"]}
        import bench_configs
        configs = {}
        # every index also at k = 1000, the reference's default n_docs (ric/conf/default.yaml:84)
        for name, kw in (("flat_10M_batch1024", dict(which="flat", n=10_000_000, steps=5, check=4, small_batches=False, extra_ks=(1000,))),
                         # BASELINE config 2 as written ("exact L2"): the same 10M through the L2 metric
                         ("flat_10M_batch1024_L2", dict(which="flat", n=10_000_000, steps=5, check=4, small_batches=False, metric="l2")),
                         ("ivfflat_100M_nlist4096_nprobe32", dict(which="ivfflat", n=100_000_000, steps=5, check=2, extra_ks=(1000,))),
                         # the reference's own shipped operating points: ric/conf/example_config.yaml:70-76 (IVFFlat, ncentroids
                         # 2048, probe 128; 20M vectors here to bound the run) and ric/conf/ivf_pq.yaml:64-78 (IVFPQ, M 16,
                         # ncentroids 8192, probe 512, n_docs 1000)
                         ("ivfflat_20M_nlist2048_nprobe128", dict(which="ivfflat", n=20_000_000, nlist=2048, nprobe=128, steps=3, check=2, extra_ks=(1000,))),
                         ("ivfpq_100M_M16_nlist8192_nprobe512", dict(which="ivfpq_ref"))):
            t0 = time.time()
            try:
                if kw["which"] == "ivfpq_ref":
                    configs[name] = bench_configs.measure_ivfpq(100_000_000, 16, 8192, 512, ks=(10, 1000), steps=3, check=2)
                else:
                    configs[name] = bench_configs.measure(**kw)
            except Exception as e:   # a config that cannot run (e.g. a smaller GPU) must not lose the headline line
                configs[name] = {"error": repr(e)}
            log(f"config {name}: {time.time() - t0:.1f}s -> {json.dumps(configs[name])[:300]}")
            gc.collect()

    if rank == 0:
        traffic, traffic_note, mfma_busy = load_pmc_traffic(n_total, world, kernel)
        # what the counters say the dominant kernel is bound by.  k_pq_scan_sl8 (round 6): 0.6-0.8 instructions per SIMD
        # issue turn, LDS array a third busy, matrix pipe a third busy — none saturated — while its L2-miss traffic (codes + table slices +
        # sibling re-reads) runs at 5.5-5.8 TB/s, nine tenths of what this chip delivers into LDS (6.2 TB/s measured): memory bound.
        # k_pq_scan_rot (rounds 2-5): CU bound — one instruction per issue turn on every SIMD (profiles/r05_scan_plateau.md section 5).
        bound_note = ("memory: L2-miss traffic (roofline.traffic) at ~0.9 of the measured 6.2 TB/s fill rate; SQ issue 0.6-0.8 per turn, LDS ~0.35, "
                      "MFMA ~0.36 busy (profiles/r06_sliced_scan.md); a build without the look-ups moves its requests at the same ~5.5 TB/s, "
                      "more loads in flight only queue (profiles/r06_large_k_flat.md 2)" if kernel == "k_pq_scan_sl8" else
                      "cu (lds gather issue): one instruction per 4-clock issue turn on every SIMD, LDS 0.61, MFMA 0.30 busy "
                      "(profiles/r05_scan_plateau.md)" if kernel == "k_pq_scan_rot" else "lds bank conflicts (granule layout)")
        res = {
            "metric": "queries/sec + recall@10, 100M x 768 IVF-PQ nprobe=32 batch=1024",
            "value": round(args.steps * nq / elapsed, 2),
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None,
            "dtype": "u8 codes; i8-table integer scan (MFMA-i8 adder tree), exact f32-table re-rank (certified)",
            "data": "synthetic",
            "recall_at_10": recall,
            "first_timed_batch_sha256": result_sha,
            "recall_low_noise_queries": recall_low,
            "recall_informative": recall2,
            "config": {"workload": f"{n_total}x{D} IVF-PQ M={args.m} nbits=8 nlist={args.nlist} nprobe={args.nprobe} "
                                   f"batch={nq} k={k}, inner product, by_residual",
                       "vectors_per_gpu": n_local, "code_layout": {2: "sliced", 1: "rotated"}.get(layout, "granule"),
                       "parallelism": (f"index sharded by inverted lists (l % {world} == rank) over {world} GPU(s)" if list_shards
                                       else f"index sharded by id range over {world} GPU(s)"),
                       "dist_backend": (args.dist_backend if world > 1 else None)},
            "roofline": {"bound": "hbm" if kernel != "k_pq_scan_rot" else "cu (lds gather issue)", "bound_by_counters": bound_note, "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": round(ms_per_launch, 4),
                         "launches_per_step": launches_per_step,
                         "traffic_gbs": (round(traffic / sec / 1e9, 2) if traffic and sec > 0 else None),
                         "traffic_over_algorithmic": (round(traffic / alg_bytes, 3) if traffic and alg_bytes > 0 else None),
                         "traffic_note": traffic_note,
                         "mfma_busy": (round(mfma_busy, 4) if mfma_busy else None),      # north_star's MFMA-busy counter: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), same stamped PMC session
                         "lds": {"achieved": round(lds_bytes / sec / 1e9, 1) if sec > 0 else None, "peak": round(lds_peak, 1), "unit": "GB/s",
                                 "frac": round(lds_bytes / sec / 1e9 / lds_peak, 4) if sec > 0 else None,
                                 "queries_per_gather": gq,
                                 "gather_padding_ratio": (round(scanned_group * gq / scanned, 4) if scanned > 0 else None),
                                 "note": f"{gather_bytes}-byte table gathers ({'ds_read_b64' if gq == 8 else 'ds_read_b32'}): one per lane per (vector, "
                                         f"sub-quantiser, group of <= {gq} queries); peak = {256 if gq == 8 else 128} B/clk/CU x 256 CUs x 2.4 GHz "
                                         "(MI355X_MICROARCH.md, LDS); gather_padding_ratio = group slots / (vector, query) pairs: the share of "
                                         "every gather that serves an empty query slot"},
                         "logical": {"bytes_per_launch": scanned * args.m / launches_per_step,
                                     "effective_GBs": round(scanned * args.m / launches_per_step / sec / 1e9, 1) if sec > 0 else None,
                                     "note": "SURVEY 8(d): sum over (query, probed list) of len*M; an EFFECTIVE rate (one HBM read serves "
                                             "every query group of the list), not a fraction of any peak"},
                         "note": "achieved = M bytes x vectors of every inverted list probed by at least one query of the batch (the bytes a "
                                 "list-major scan must pull from HBM once) / HIP-event duration of the scan launch on the library "
                                 "stream, rank 0; traffic = PMC FETCH_SIZE x 2 of the same kernel (profiles/pmc_traffic.json, refused when "
                                 "the kernel sources changed since it was measured)"},
            "pcie_inclusive": pcie,
            "one_call_all_queries": one_call,
            "list_length_histogram": hist,
            "settle_batches": SETTLE,
            "stage_ms_per_step": stage_ms,
            "stage_ms_note": ("stages marked inside the timed region (profile 1)" if stages_in_timed_region else
                              "the timed region carries HIP events around the scan launch only (profile -1: roofline.ms_per_launch); this per-stage breakdown is a "
                              "second pass over the same batches with an event per stage (each ~5 us of stream time), outside the timed region"),
            "certificate_fallback_fraction": fallbacks,
            "filter_survivors_per_query": {"mean": round(cand_keys / max(1, nq), 1), "max": cand_keys_max},
            "ab_exact_kernels_same_process": ab,
            "reference_n_docs_on_this_index": ops,
            "other_distributions": dist_legs,
            "configs": configs,
            "cpu_baseline": cpu,
            "cpu_parity_ids_and_scores_bit_exact": parity,
            "faiss_baseline": faiss_res,
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command line under torch.distributed.run with one rank
    per GPU (the form the driver uses for N > 1), on a free loopback port; the ranks' stdout/stderr pass through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]   # "--": bench flags such as --n are not the launcher's
    log("self-launch:", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def kernel_source_hash():
    """sha256 over the sources that decide the scan kernel's HBM traffic (kernel, grouping, tile choice)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("k_pq_rot.hip", "k_pq.hip", "k_select.hip", "api_search.hip", "rsx_internal.h"):
        h.update(open(os.path.join(REPO, "retrieval-scaling_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def load_pmc_traffic(n_total, world, kernel):
    """HBM bytes per scan launch from the committed rocprofv3 --pmc FETCH_SIZE pass (tools/update_pmc_traffic.py writes the
    file with the hash of the sources it measured) — null when it does not describe THIS build / workload."""
    tpath = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, "profiles/pmc_traffic.json missing", None
    if t.get("n") != n_total or t.get("n_gpus") != world:
        return None, "pmc_traffic.json describes another workload", None
    if t.get("kernel") != kernel:
        return None, f"pmc_traffic.json was measured on {t.get('kernel')}, this run used {kernel}", None
    if t.get("source_sha256") != kernel_source_hash():
        return None, "pmc_traffic.json is stale: the kernel sources changed since the PMC pass", None
    return t.get("hbm_bytes_per_launch"), f"PMC pass of {t.get('date', '?')}, FETCH_SIZE x 2 (gfx950 correction), same sources", t.get("mfma_busy_frac")


if __name__ == "__main__":
    main()
