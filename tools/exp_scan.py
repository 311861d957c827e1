#!/usr/bin/env python3
"""Experiment driver (not the bench): build the headline 100M x 768 IVF-PQ index ONCE, then time rsx_search under a list of
engine-parameter settings in the same process, checking every setting's results against the first one's bit for bit.

  python tools/exp_scan.py --set pq_pace=0 --set pq_pace=1 --set pq_pace=2,scan_chunk=16384
"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "retrieval-scaling_amd")); sys.path.insert(0, REPO)
import numpy as np

D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--m", type=int, default=96)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2, help="every setting is timed this many times, interleaved")
    ap.add_argument("--set", action="append", default=[], help="comma-separated name=value engine parameters; k=.. / nprobe=.. / batch=.. are search arguments")
    ap.add_argument("--layouts", default="", help="comma-separated pq_layout values: one index per layout over the same vectors (same centroids and codebooks), "
                    "every setting is timed on each, interleaved, and all results are compared with the first's")
    args = ap.parse_args()
    import torch, rsx
    dev = torch.device("cuda", 0)
    layouts = [int(v) for v in args.layouts.split(",") if v != ""] or [None]
    ixs = []
    for lay in layouts:
        jx = rsx.IndexIVFPQ(rsx.IndexFlatIP(D), D, args.nlist, args.m, 8, rsx.METRIC_INNER_PRODUCT, device=0)
        if lay is not None:
            jx.set_param("pq_layout", float(lay))
        ixs.append(jx)
    ix = ixs[0]
    nt = min(args.n, 256 * args.nlist)
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    stride = max(1, args.n // nt)
    for b in range(0, nt, 4096):
        nb = min(4096, nt - b)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, (b * stride) % max(1, args.n - nb), nb, out=xt[b:b + nb])
    t0 = time.time(); ix.train(xt); del xt
    for jx in ixs[1:]:
        jx.set_centroids(ix.get_centroids()); jx.set_codebooks(ix.get_codebooks())
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, args.n, 1_000_000):
        nb = min(1_000_000, args.n - c0)
        rsx.synth_vectors(D, NC, SC, SX, 0.5, c0, nb, out=buf[:nb])
        for jx in ixs:
            jx.add(buf[:nb])
    del buf
    print(f"[exp] build {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    nsteps = args.warmup + args.steps
    maxb = max([args.batch] + [int(dict(kv.split("=") for kv in s.split(",")).get("batch", args.batch)) for s in args.set if s])
    Q = torch.empty((nsteps * maxb, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NC, SC, SX, 0.5, args.n, SQ, 0.1, 0, nsteps * maxb, out=Q)
    settings = args.set or [""]
    ref = {}
    out = []
    for rnd in range(args.rounds):
      for s in settings:
        for lay, ix in zip(layouts, ixs):
            kv = dict(x.split("=") for x in s.split(",")) if s else {}
            k = int(kv.pop("k", args.k)); nprobe = int(kv.pop("nprobe", args.nprobe)); nq = int(kv.pop("batch", args.batch))
            ix.nprobe = nprobe
            for name, v in kv.items():
                ix.set_param(name, float(v))
            for i in range(args.warmup):
                ix.search(Q[i * nq:(i + 1) * nq], k)
            ix.set_param("profile", 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(args.warmup, nsteps):
                Dd, Ii = ix.search(Q[i * nq:(i + 1) * nq], k)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            st = {x: round(ix.get_timing(x) / args.steps, 4) for x in ("coarse", "select_probe", "lut8", "group", "scan0", "scan", "select", "finalize", "total")}
            fbst = {x: round(ix.get_timing("fb_" + x), 3) for x in ("convert", "coarse", "select_probe", "lut", "scan", "select", "finalize", "total")}
            fbl = ix.get_timing("fb_scan_launches")
            fb = ix.get_timing("fallback_queries")
            fbo = ix.get_timing("fallback_overflow_queries")
            sec = ix.get_timing("second_chance_queries")
            ix.set_param("profile", 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(args.warmup, nsteps):
                ix.search(Q[i * nq:(i + 1) * nq], k)
            torch.cuda.synchronize(); el0 = time.perf_counter() - t0
            ix.set_param("profile", 2); ix.search(Q[:nq], k)
            ck, ckm = ix.get_timing("cand_keys") / nq, ix.get_timing("cand_keys_max")
            ix.set_param("profile", 0)
            key = (k, nprobe, nq)
            Dn, In = Dd.cpu().numpy(), Ii.cpu().numpy()
            same = None
            if key in ref:
                same = bool(np.array_equal(ref[key][0], Dn) and np.array_equal(ref[key][1], In))
            else:
                ref[key] = (Dn, In)
            for name in kv:      # back to defaults for the next setting
                ix.set_param(name, {"pq_pace": float(128 | (4 << 12)), "pq_pre_rows": 4096.0, "scan_chunk": 0.0, "pq_prune": 0.0, "pq_fast_kp": 0.0, "overlap": 0.0, "lut_tiled": 2.0, "pq_pre_mult": 80.0, "pq_pre_max": 16384.0}.get(name, 1.0))
            r = {"set": s, "layout": lay, "round": rnd, "qps": round(args.steps * nq / el, 1), "ms_per_step": round(el / args.steps * 1e3, 4), "ms_per_step_unprofiled": round(el0 / args.steps * 1e3, 4), "stages": st,
                 "fallback_queries": fb, "fb_stage_ms_total": fbst, "fb_launches": fbl, "fallback_overflow": fbo, "second_chance_queries": sec, "cand_mean": round(ck, 1), "cand_max": ckm, "same_as_first": same}
            print(json.dumps(r), flush=True); out.append(r)


if __name__ == "__main__":
    main()
