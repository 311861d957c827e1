"""GPU: size-independent properties at larger sizes than the oracle can brute-force, and the
real Indexer facade over pickle shards."""
import os
import pickle

import numpy as np
import pytest

from util import assert_same_results

pytestmark = pytest.mark.gpu


def test_properties_200k(gpu, orc):
    import torch
    d, n, nlist, M, nq, k = 768, 200_000, 256, 96, 256, 10
    x = torch.empty((n, d), dtype=torch.float16, device="cuda")
    gpu.synth_vectors(d, 256, 1234, 10000, 0.5, 0, n, out=x)
    q = torch.empty((nq, d), dtype=torch.float16, device="cuda")
    gpu.synth_queries(d, 256, 1234, 10000, 0.5, n, 999, 0.1, 0, nq, out=q)
    torch.cuda.synchronize()
    flat = gpu.IndexFlatIP(d); flat.add(x)
    Dg, Ig = flat.search(q, k)
    Dg, Ig = Dg.cpu().numpy(), Ig.cpu().numpy()
    # spot-check the exact index against the oracle on a sample of queries
    xs = x.cpu().numpy()
    Dr, Ir = orc.flat_search(q[:8].cpu().numpy().astype(np.float32), xs.astype(np.float32), k, 0)
    assert_same_results(Dg[:8], Ig[:8], Dr, Ir, "flat 200k")
    flat.set_param("flat_filter", 0)        # score buffer per chunk instead of the filtered single launch
    Dn, In = flat.search(q, k)
    flat.set_param("flat_filter", 1)
    assert_same_results(Dn.cpu().numpy(), In.cpu().numpy(), Dg, Ig, "flat filtered vs per-chunk")
    Dk2, Ik2 = flat.search(q, 100)          # larger k through the filtered path
    flat.set_param("flat_filter", 0)
    Dk3, Ik3 = flat.search(q, 100)
    flat.set_param("flat_filter", 1)
    assert torch.equal(Dk2, Dk3) and torch.equal(Ik2, Ik3)
    # sharded Flat: two half indexes + merge kernel == one index (exact)
    h0, h1 = gpu.IndexFlatIP(d), gpu.IndexFlatIP(d)
    h0.add(x[: n // 2]); h1.add(x[n // 2:])
    D0, I0 = h0.search(q, k); D1, I1 = h1.search(q, k)
    Dm, Im = gpu.merge_topk(torch.stack([D0, D1]), torch.stack([I0, I1 + n // 2]))
    assert_same_results(Dm.cpu().numpy(), Im.cpu().numpy(), Dg, Ig, "sharded flat")
    # IVF-Flat: nprobe = nlist is exhaustive; recall grows with nprobe
    ivf = gpu.IndexIVFFlat(None, d, nlist, 0)
    ivf.train(x[:65536]); ivf.add(x)
    ivf.nprobe = nlist
    D, I = ivf.search(q, k)
    assert_same_results(D.cpu().numpy(), I.cpu().numpy(), Dg, Ig, "ivfflat exhaustive")
    rec = []
    for nprobe in (1, 8, 64):
        ivf.nprobe = nprobe
        _, I = ivf.search(q, k)
        I = I.cpu().numpy()
        rec.append(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(I, Ig)]))
    assert rec[0] <= rec[1] <= rec[2] and rec[2] > 0.9, rec
    # IVF-PQ: idempotent, batch-invariant, and identical to the oracle's scan on a sample
    pq = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    pq.set_centroids(ivf.get_centroids())
    pq.train(x[:65536]) if False else None
    cen = ivf.get_centroids()
    a, _ = orc.assign_ip(cen, xs[:20000].astype(np.float32))
    cb = orc.pq_train(orc.residuals(cen, xs[:20000].astype(np.float32), a)[:4096], M, 2, 1234)
    pq.set_codebooks(cb)
    pq.add(x)
    pq.nprobe = 16
    D1, I1 = pq.search(q, k)
    D2, I2 = pq.search(q, k)
    assert torch.equal(D1, D2) and torch.equal(I1, I2)
    pq.set_param("query_batch", 100)
    D3, I3 = pq.search(q, k)
    assert torch.equal(D1, D3) and torch.equal(I1, I3)
    # fast scan (default) == exact list-major == exact per-pair kernels at a size where selection prunes
    pq.set_param("query_batch", 1024)
    for sk in (2, 1):
        pq.set_param("scan_kernel", sk)
        Dk, Ik = pq.search(q, k)
        assert torch.equal(D1, Dk) and torch.equal(I1, Ik), f"scan_kernel={sk}"
    pq.set_param("scan_kernel", 0)
    for kk in (1, 100):
        pq.nprobe = 64
        Df, If = pq.search(q, kk)
        pq.set_param("pq_fast", 0)
        Dx, Ix = pq.search(q, kk)
        pq.set_param("pq_fast", 1)
        assert torch.equal(Df, Dx) and torch.equal(If, Ix), f"fast vs exact k={kk}"
    pq.nprobe = 16
    pq.set_param("pq_filter", 0)
    Du, Iu = pq.search(q, k)
    pq.set_param("pq_filter", 1)
    assert torch.equal(D1, Du) and torch.equal(I1, Iu), "filtered vs unfiltered fast scan"
    # oracle on the exported lists for 4 queries
    lists = [pq.get_list(l) for l in range(nlist)]
    off = np.zeros(nlist + 1, np.int64); np.cumsum([len(i) for _, i in lists], out=off[1:])

    class LM: pass
    lm = LM(); lm.list_off = off
    lm.payload = np.concatenate([c for c, _ in lists]); lm.ids = np.concatenate([i for _, i in lists])
    Dr, Ir = orc.ivfpq_search(cen, cb, lm, q[:4].cpu().numpy().astype(np.float32), 16, k)
    assert_same_results(D1[:4].cpu().numpy(), I1[:4].cpu().numpy(), Dr, Ir, "ivfpq 200k vs oracle")


class NS(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


@pytest.mark.parametrize("index_type", ["Flat", "IVFFlat", "IVFPQ"])
def test_indexer_facade_on_gpu(gpu, orc, tmp_path, index_type):
    """Config 1 shape (a few thousand fp16 x 768 passages, k = 3) through Indexer(cfg).search."""
    from src.indicies.base import Indexer
    tmp = str(tmp_path)
    os.makedirs(os.path.join(tmp, "emb")); os.makedirs(os.path.join(tmp, "psg"))
    d, per = 768, 1800
    embs = []
    for s in range(2):
        e = orc.synth_vectors(d, 16, 11, 100 + s, 0.5, 0, per)
        embs.append(e)
        with open(os.path.join(tmp, "emb", f"passages_{s:02d}.pkl"), "wb") as f:
            pickle.dump((list(range(per)), e), f)
        with open(os.path.join(tmp, "psg", f"raw_passages-{s}-of-2.pkl"), "wb") as f:
            pickle.dump([{"text": f"s{s}c{c}"} for c in range(per)], f)
    cfg = NS(datastore=NS(embedding=NS(embedding_dir=os.path.join(tmp, "emb"), prefix="passages",
                                       passages_dir=os.path.join(tmp, "psg")),
                          index=NS(index_type=index_type, index_shard_ids=[0, 1], projection_size=d,
                                   sample_train_size=3000, ncentroids=16, probe=16, n_subquantizers=96, n_bits=8)))
    ix = Indexer(cfg)
    allx = np.concatenate(embs, 0)
    q = orc.synth_queries(d, 16, 11, 100, 0.5, per, 7, 0.1, 0, 300)
    scores, passages, db_ids = ix.search(q, k=3)
    D, I = orc.flat_search(q.astype(np.float32), allx.astype(np.float32), 3, 0)
    if index_type != "IVFPQ":       # probe = ncentroids -> exhaustive -> identical to brute force
        assert db_ids == [[[int(i) // per, int(i) % per] for i in row] for row in I]
        assert scores == D.tolist()
        assert passages[0][0] == f"s{I[0, 0] // per}c{I[0, 0] % per}"
    else:
        got = np.array([[s_ * per + c for s_, c in row] for row in db_ids])
        rec = np.mean([len(set(a.tolist()) & set(b.tolist())) / 3 for a, b in zip(got, I)])
        assert rec > 0.3
    ix2 = Indexer(cfg)              # reload from the files written above
    s2, p2, d2 = ix2.search(q, k=3)
    assert d2 == db_ids and s2 == scores and p2 == passages


def _oracle_lists(orc, ix, need, nlist):
    off = np.zeros(nlist + 1, np.int64); lens = np.zeros(nlist, np.int64); pay, ids = [], []
    for l in need:
        v, i = ix.get_list(int(l)); pay.append(v); ids.append(i); lens[l] = len(i)
    np.cumsum(lens, out=off[1:])

    class LM: pass
    lm = LM(); lm.list_off = off; lm.payload = np.concatenate(pay); lm.ids = np.concatenate(ids)
    return lm


@pytest.mark.parametrize("metric", [0, 1])
def test_flat_batch_1024(gpu, orc, metric):
    """BASELINE config 2's batch shape (1024 queries through the 256 x 256 MFMA tiles and the filtered second pass) at 1M
    vectors, inner product and — config 2 as written, "exact L2" — the L2 metric: a sample of queries against the oracle,
    and the whole batch against itself in four sub-batches."""
    import torch
    d, n, nq, k = 768, 1_000_000, 1024, 10
    x = torch.empty((n, d), dtype=torch.float16, device="cuda")
    gpu.synth_vectors(d, 4096, 1234, 10000, 0.5, 0, n, out=x)
    q = torch.empty((nq, d), dtype=torch.float16, device="cuda")
    gpu.synth_queries(d, 4096, 1234, 10000, 0.5, n, 999, 0.1, 0, nq, out=q)
    ix = gpu.IndexFlat(d, metric); ix.add(x)
    ix.set_param("profile", 1)
    D, I = ix.search(q, k)
    assert ix.get_timing("fallback_queries") == 0          # fp16 data: the MFMA scan certifies every query (both metrics)
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    sample = [0, 255, 256, 511, 777, 1023]
    Dr, Ir = orc.flat_search(q[sample].cpu().numpy().astype(np.float32), x.cpu().numpy().astype(np.float32), k, metric)
    assert_same_results(Dn[sample], In[sample], Dr, Ir, f"flat 1M batch 1024 metric {metric} vs oracle")
    for s in range(0, nq, 256):
        Ds, Is = ix.search(q[s:s + 256], k)
        assert torch.equal(Ds, D[s:s + 256]) and torch.equal(Is, I[s:s + 256]), "batch decomposition must be invisible"


@pytest.mark.parametrize("kind", ["ivfflat", "ivfpq"])
def test_ivf_nlist4096_nprobe32_batch_1024(gpu, orc, kind):
    """BASELINE configs 3 / 4's index parameters (nlist 4096, nprobe 32, batch 1024; IVF-PQ M = 96 on the rotated layout) at
    1.5M vectors: a sample of queries against the oracle on the exported lists, batch-split invariance, and the probe
    selection against the oracle's coarse quantiser."""
    import torch
    d, n, nlist, nprobe, nq, k = 768, 1_500_000, 4096, 32, 1024, 10
    x = torch.empty((n, d), dtype=torch.float16, device="cuda")
    gpu.synth_vectors(d, 4096, 1234, 10000, 0.5, 0, n, out=x)
    q = torch.empty((nq, d), dtype=torch.float16, device="cuda")
    gpu.synth_queries(d, 4096, 1234, 10000, 0.5, n, 999, 0.1, 0, nq, out=q)
    ix = gpu.IndexIVFFlat(None, d, nlist, 0) if kind == "ivfflat" else gpu.IndexIVFPQ(None, d, nlist, 96, 8, 0)
    ix.train(x[:524288]); ix.add(x); ix.nprobe = nprobe
    ls = ix.list_sizes()
    assert ls.sum() == n and ls.max() > 2 * np.median(ls)          # k-means lists are NOT balanced
    D, I = ix.search(q, k)
    sample = [0, 256, 513, 1023]
    qs = q[sample].cpu().numpy().astype(np.float32)
    cen = ix.get_centroids()
    pid, _ = orc.coarse_probe(cen, qs, nprobe)
    lm = _oracle_lists(orc, ix, np.unique(pid), nlist)
    if kind == "ivfflat":
        Dr, Ir = orc.ivfflat_search(0, cen, lm, qs, nprobe, k)
    else:
        Dr, Ir = orc.ivfpq_search(cen, ix.get_codebooks(), lm, qs, nprobe, k)
    assert_same_results(D[sample].cpu().numpy(), I[sample].cpu().numpy(), Dr, Ir, f"{kind} nlist 4096 nprobe 32 vs oracle")
    for s in range(0, nq, 512):
        Ds, Is = ix.search(q[s:s + 512], k)
        assert torch.equal(Ds, D[s:s + 512]) and torch.equal(Is, I[s:s + 512]), "batch decomposition must be invisible"
    if kind == "ivfpq":
        ix.set_param("pq_prune", 1)
        Dp, Ip = ix.search(q, k)
        assert torch.equal(Dp, D) and torch.equal(Ip, I), "pair pruning is exact"


def test_m16_reference_shape_mid_scale(gpu, orc):
    """The reference's shipped IVF-PQ shape (M = 16, many probes per query: ric/conf/ivf_pq.yaml:64-78) at a size where the round-4
    M = 16 scan (k_pq_scan_rot16: 16 queries per work item, four records, work stealing across XCD ranges, lists of several tiles
    when scan_chunk shrinks them) runs thousands of work items: every query of the batch against the exact list-major kernel,
    a sample against the CPU oracle, for the reference's n_docs too; no exact re-run may be needed."""
    import torch
    d, n, nlist, M, nq, nprobe = 768, 2_000_000, 256, 16, 256, 64
    dev = torch.device("cuda", 0)
    x = torch.empty((n, d), dtype=torch.float16, device=dev)
    gpu.synth_vectors(d, 512, 1234, 10000, 0.5, 0, n, out=x)
    q = torch.empty((nq, d), dtype=torch.float16, device=dev)
    gpu.synth_queries(d, 512, 1234, 10000, 0.5, n, 999, 0.1, 0, nq, out=q)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    assert ix._get("pq_layout") == 1
    ix.train(x[:65536]); ix.add(x); ix.nprobe = nprobe
    del x
    qs = q[:6].cpu().numpy().astype(np.float32)
    cen, cb = ix.get_centroids(), ix.get_codebooks()
    pid, _ = orc.coarse_probe(cen, qs, nprobe)
    need = np.unique(pid[pid >= 0])
    ls = ix.list_sizes()
    lens = np.zeros(nlist, np.int64); lens[need] = ls[need]
    off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])

    class LM: pass
    lm = LM(); lm.list_off = off
    lm.payload = np.empty((int(off[-1]), M), np.uint8); lm.ids = np.empty(int(off[-1]), np.int64)
    for l in need:
        c, i = ix.get_list(int(l)); lm.payload[off[l]:off[l + 1]] = c; lm.ids[off[l]:off[l + 1]] = i
    for k in (10, 1000):
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q, k)
        ix.set_param("scan_kernel", 0)
        for chunk in (0, 2048):          # 2048: every list becomes several tiles
            ix.set_param("scan_chunk", chunk); ix.set_param("profile", 1)
            D, I = ix.search(q, k)
            assert torch.equal(I, Ie) and torch.equal(D, De), f"k={k} scan_chunk={chunk}: fast scan vs exact kernel"
            assert ix.get_timing("fallback_queries") == 0, f"k={k} scan_chunk={chunk}"
        ix.set_param("scan_chunk", 0); ix.set_param("profile", 0)
        Do, Io = orc.ivfpq_search(cen, cb, lm, qs, nprobe, k)
        assert np.array_equal(Io, I[:6].cpu().numpy()) and np.array_equal(Do, D[:6].cpu().numpy()), f"k={k}: oracle sample"


@pytest.mark.parametrize("metric", [0, 1])
def test_flat_staged_filter_at_the_reference_n_docs(gpu, orc, metric):
    """Flat at the reference's n_docs (ric/conf/default.yaml:70,84: Flat, n_docs 1000) behind the staged threshold: the rows
    behind the threshold phase are scanned in stages of growing row ranges, each filtered by the running K'-th key and followed
    by a selection that tightens it (api_search.hip: search_batch).  Any number of stages and any threshold-phase size must
    return the oracle's ids and scores; the default must take more than one stage here."""
    import torch
    d, n, nq = 128, 1_000_000, 160
    x = torch.empty((n, d), dtype=torch.float16, device="cuda")
    gpu.synth_vectors(d, 256, 1234, 10000, 0.5, 0, n, out=x)
    q = torch.empty((nq, d), dtype=torch.float16, device="cuda")
    gpu.synth_queries(d, 256, 1234, 10000, 0.5, n, 999, 0.1, 0, nq, out=q)
    ix = gpu.IndexFlat(d, metric); ix.add(x)
    sample = [0, 63, 64, 159]
    xs = x.cpu().numpy().astype(np.float32)
    for k in (1000, 100):
        Dr, Ir = orc.flat_search(q[sample].cpu().numpy().astype(np.float32), xs, k, metric)
        ref = None
        for mult, stages, unit in ((16, 0, 0), (32, 0, 65536), (64, 1, 4096), (1, 4, 0), (16, 2, 65536), (160, 3, 1000), (4, 0, 16384)):
            ix.set_param("flat_pre_mult", mult); ix.set_param("flat_stages", stages); ix.set_param("flat_pre_unit", unit); ix.set_param("profile", 1)
            D, I = ix.search(q, k)
            assert ix.get_timing("fallback_queries") == 0 and ix.get_timing("flat_filter_overflows") == 0
            Dn, In = D.cpu().numpy(), I.cpu().numpy()
            assert_same_results(Dn[sample], In[sample], Dr, Ir, f"flat k={k} metric {metric} pre_mult={mult} stages={stages} unit={unit} vs oracle")
            if ref is None:
                ref = (Dn, In)
            assert np.array_equal(ref[1], In) and np.array_equal(ref[0], Dn), f"k={k} pre_mult={mult} stages={stages} unit={unit}: staging must be invisible"
        ix.set_param("flat_pre_mult", 16); ix.set_param("flat_stages", 0); ix.set_param("flat_pre_unit", 0); ix.set_param("profile", 0)


@pytest.mark.parametrize("metric", [0, 1])
def test_ivfflat_filter_at_the_reference_n_docs(gpu, orc, metric):
    """IVF-Flat at the reference's n_docs (ric/conf: n_docs 1000) behind the in-kernel filter: the threshold comes from a sample of
    the first rows of the query's closest list(s) — 1, 2, 4 or 8 of them (ivf_pre_lists) — and lists of very different lengths
    (some shorter than one scan chunk or than K', some of 20+ chunks) must neither lose a result nor change one: the oracle's ids,
    the scores within the fp16-scan tolerance of the other IVF-Flat tests, identical results across the settings and with the
    unfiltered scan."""
    rng = np.random.RandomState(5)
    d, nlist, n, nq, nprobe = 128, 24, 260_000, 130, 12
    cen = rng.randn(nlist, d).astype(np.float32)
    pl = rng.dirichlet(np.full(nlist, 0.6))                 # skewed list sizes: a few hundred to tens of thousands of rows
    x = (cen[rng.choice(nlist, n, p=pl)] + 0.5 * rng.randn(n, d)).astype(np.float16)
    qf = (x[rng.randint(0, n, nq)].astype(np.float32) + 0.1 * rng.randn(nq, d)).astype(np.float32)
    xf = x.astype(np.float32)
    a, _ = orc.assign_ip(cen, xf)
    lm = orc.ListMajor(a, np.arange(n), xf, nlist)
    ix = gpu.IndexIVFFlat(None, d, nlist, metric)
    ix.set_centroids(cen); ix.add(x); ix.nprobe = nprobe
    ls = ix.list_sizes()
    assert ls.min() < 1024 and ls.max() > 16 * 1024, "the test wants short and long lists"
    for k in (1000, 300):
        Dr, Ir = orc.ivfflat_search(metric, cen, lm, qf[:6], nprobe, k)
        ix.set_param("ivf_filter", 0)
        Du, Iu = ix.search(qf, k)
        assert np.array_equal(Iu[:6], Ir), f"metric={metric} k={k} unfiltered vs oracle"
        assert np.allclose(Du[:6], Dr, rtol=0, atol=max(1e-30, np.abs(Dr[np.isfinite(Dr)]).max() * 2 ** -22))
        ix.set_param("ivf_filter", 2)
        for pre_lists in (0, 1, 2, 4):
            ix.set_param("ivf_pre_lists", pre_lists); ix.set_param("profile", 1)
            D, I = ix.search(qf, k)
            # (round 6: one or two overflowed candidate rows no longer send the batch through the unfiltered scan — those queries join the
            #  exact re-run; every fallback here must be such an overflow, never a failed certificate)
            assert ix.get_timing("fallback_queries") == ix.get_timing("ivf_filter_overflow_queries") <= 2
            assert np.array_equal(Iu, I) and np.array_equal(Du, D), f"metric={metric} k={k} ivf_pre_lists={pre_lists}: the filter must be invisible"
        ix.set_param("profile", 0); ix.set_param("ivf_pre_lists", 0)


def test_ivfflat_filter_overflow_of_a_few_queries_is_settled_per_query(gpu, orc):
    """IVF-Flat behind the in-kernel filter: the threshold of a query is the K'-th key of a sample of the FIRST rows of its closest
    list.  Two adversarial queries whose closest list starts with 8192 rows that score badly for them (and goes on with 22 000 that score
    well) get a useless threshold and overflow their candidate rows.  Round 5 sent the whole batch through the unfiltered scan for
    that; now k_finalize flags the overflowed rows' queries and they alone join the exact re-run (api_search.hip, k_select.hip:
    certify_rows).  All 130 queries: the oracle's ids and scores."""
    rng = np.random.RandomState(17)
    d, nlist, per, nq, k = 128, 4, 30_000, 130, 10
    cen = (3.0 * rng.randn(nlist, d)).astype(np.float32)
    u = rng.randn(d).astype(np.float32)
    for c in cen:                                   # u orthogonal to every centroid: +-u does not move a row to another list
        u -= (u @ c) / (c @ c) * c
    for _ in range(3):
        for c in cen:
            u -= (u @ c) / (c @ c) * c
    u /= np.linalg.norm(u)
    lab = np.repeat(np.arange(nlist), per)
    x = cen[lab] + 0.3 * rng.randn(nlist * per, d).astype(np.float32)
    bad = 8192
    x[:bad] -= 8.0 * u[None, :]                                                           # list 0, first rows: far below for a query along +u
    x[bad:per] += (4.0 + 4.0 * rng.rand(per - bad, 1).astype(np.float32)) * u[None, :]    # list 0, the rest: far above
    x = x.astype(np.float16)
    src = per + rng.randint(0, (nlist - 1) * per, nq)                              # ordinary queries: near rows of lists 1 .. 3
    q = (x[src].astype(np.float32) + 0.05 * rng.randn(nq, d)).astype(np.float32)
    q[5] = cen[0] + 6.0 * u; q[77] = cen[0] + 5.0 * u                               # the two adversarial ones
    q = q.astype(np.float16)
    a, _ = orc.assign_ip(cen, x.astype(np.float32))
    assert np.array_equal(a, lab), "the construction keeps every row in its list"
    ix = gpu.IndexIVFFlat(None, d, nlist, 0)
    ix.set_centroids(cen); ix.add(x); ix.nprobe = nlist
    ix.set_param("ivf_filter", 2); ix.set_param("profile", 1)
    D, I = ix.search(q, k)
    Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, 0)       # nprobe = nlist: exhaustive
    assert_same_results(D, I, Dr, Ir, "IVF-Flat, two queries with overflowing candidate rows")
    assert ix.get_timing("ivf_filter_overflow_queries") == 2, "exactly the two adversarial queries overflow"
    assert ix.get_timing("fallback_queries") >= 2
