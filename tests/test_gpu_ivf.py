"""GPU: IVF-Flat and IVF-PQ through the C-ABI against the oracle and the golden fixtures."""
import os

import numpy as np
import pytest

from util import assert_same_results, load_golden, regen_gpu, sha

pytestmark = pytest.mark.gpu


def build_ivfflat(gpu, g, x, metric=0):
    ix = gpu.IndexIVFFlat(gpu.IndexFlatIP(g["d"]), g["d"], g["nlist"], metric)
    ix.set_centroids(g["centroids"])
    ix.add(x)
    return ix


def test_golden_ivfflat(gpu, orc):
    g = load_golden("ivfflat_d768")
    x, q = regen_gpu(gpu, g)
    ix = build_ivfflat(gpu, g, x)
    # list membership = argmax-IP assignment of the oracle, insertion order inside each list
    a, _ = orc.assign_ip(g["centroids"], x.astype(np.float32))
    assert sha(a) == g["assign_sha"]
    for l in range(g["nlist"]):
        vecs, ids = ix.get_list(l)
        want = np.nonzero(a == l)[0]
        assert np.array_equal(ids, want), f"list {l} membership/order"
        assert np.array_equal(vecs, x[want].astype(np.float32))
    # quantizer.assign == the oracle's assignment; counting + exact reserve gives the same index
    assert np.array_equal(ix.assign(x), a)
    ix_r = gpu.IndexIVFFlat(None, g["d"], g["nlist"], 0)
    ix_r.set_centroids(g["centroids"])
    ix_r.reserve_lists(np.bincount(ix_r.assign(x), minlength=g["nlist"]))
    ix_r.add(x)
    ix_r.nprobe = g["nprobe"]
    Dr_, Ir_ = ix_r.search(q, g["k"])
    assert_same_results(Dr_, Ir_, g["D"], g["I"], "ivfflat golden (reserved build)")
    ix.nprobe = g["nprobe"]
    D, I = ix.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], "ivfflat golden")
    ix.set_param("ivf_filter", 2)          # in-kernel candidate filtering (auto-enabled only for multi-GB score rows)
    Du, Iu = ix.search(q, g["k"])
    assert_same_results(Du, Iu, g["D"], g["I"], "ivfflat golden, filtered")
    ix.set_param("ivf_filter", 0)
    ix.set_param("scan_chunk", 256)        # ... and through the VGPR-staged kernel (k_list_scan)
    Dv, Iv = ix.search(q, g["k"])
    assert_same_results(Dv, Iv, g["D"], g["I"], "ivfflat golden, k_list_scan")
    ix.set_param("scan_chunk", 0); ix.set_param("ivf_filter", 1)
    # nprobe = nlist must reproduce the exhaustive search (and hence the Flat index)
    ix.nprobe = g["nlist"]
    D, I = ix.search(q, g["k"])
    Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), g["k"], 0)
    assert_same_results(D, I, Dr, Ir, "nprobe=nlist")
    ix.nprobe = 1000                      # more probes than lists: clamped, as FAISS does
    D2, I2 = ix.search(q, g["k"])
    assert_same_results(D2, I2, Dr, Ir, "nprobe>nlist")


@pytest.mark.parametrize("name,layout", [("ivfpq_d64_m16", 1), ("ivfpq_d64_m16", 0), ("ivfpq_d768_m96", 2), ("ivfpq_d768_m96", 1), ("ivfpq_d768_m96", 0)])
def test_golden_ivfpq(gpu, orc, name, layout):
    """layout 1 = the rotated code layout (k_pq_rot.hip: conflict-free table gathers, the default for M in {16, 32, 64, 128} — M = 16,
    the reference's shipped IVF-PQ config, since round 4), layout 2 = the sliced layout (round 6, M = 96: eight queries per table
    gather, k_pq_scan_sl8), layout 0 = the granule layout (k_pq.hip); all must give the oracle's bits through every scan variant."""
    g = load_golden(name)
    x, q = regen_gpu(gpu, g)
    ix = gpu.IndexIVFPQ(gpu.IndexFlatIP(g["d"]), g["d"], g["nlist"], g["M"], 8, gpu.METRIC_INNER_PRODUCT)
    assert ix._get("pq_layout") == (2 if g["M"] == 96 else 1), "a block layout is the default for M in {16, 32, 64, 96, 128}: sliced for M = 96, else rotated"
    ix.set_param("pq_layout", layout)
    assert ix._get("pq_layout") == layout
    name = f"{name} layout={layout}"
    assert not ix.is_trained
    ix.set_centroids(g["centroids"])
    ix.set_codebooks(g["codebooks"])
    assert ix.is_trained
    ix.add(x[:1000]); ix.add(x[1000:])
    assert ix.ntotal == g["n"]
    # codes are bit-identical to ProductQuantizer::compute_code as restated by the oracle
    a, _ = orc.assign_ip(g["centroids"], x.astype(np.float32))
    codes = orc.pq_encode(g["codebooks"], orc.residuals(g["centroids"], x.astype(np.float32), a))
    assert sha(a) == g["assign_sha"] and sha(codes) == g["codes_sha"]
    for l in range(g["nlist"]):
        c, ids = ix.get_list(l)
        want = np.nonzero(a == l)[0]
        assert np.array_equal(ids, want), f"list {l} membership/order"
        assert np.array_equal(c, codes[want]), f"list {l} codes"
    ix.nprobe = g["nprobe"]
    D, I = ix.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], name)           # ids AND fp32 scores bit-exact
    # same recall@k as the CPU path at identical nprobe (north star) — trivially, since ids are equal
    Dgt, Igt = orc.flat_search(q.astype(np.float32), x.astype(np.float32), g["k"], 0)
    rec = np.mean([len(set(a_.tolist()) & set(b_.tolist())) / g["k"] for a_, b_ in zip(I, Igt)])
    assert abs(rec - g["recall"]) < 1e-12
    # one query at a time / odd batch sizes give the same answer (work decomposition is invisible)
    ix.set_param("query_batch", 5)
    D2, I2 = ix.search(q, g["k"])
    assert_same_results(D2, I2, g["D"], g["I"], name + " batched")
    D1, I1 = ix.search(q[7:8], g["k"])
    assert_same_results(D1, I1, g["D"][7:8], g["I"][7:8], name + " single")
    ix.set_param("scan_chunk", 64)          # smallest scan chunks: many work items per list
    D3, I3 = ix.search(q, g["k"])
    assert_same_results(D3, I3, g["D"], g["I"], name + " chunked")
    ix.set_param("scan_chunk", 0)
    ix.set_param("query_batch", 1024)
    # every scan implementation returns the same bits: fast (8-bit tables + certified re-rank, default),
    # exact list-major (2), exact per-pair (1)
    for sk in (1, 2):
        ix.set_param("scan_kernel", sk)
        D4, I4 = ix.search(q, g["k"])
        assert_same_results(D4, I4, g["D"], g["I"], name + f" scan_kernel={sk}")
    ix.set_param("scan_kernel", 0)
    ix.set_param("pq_fast", 0)
    D5, I5 = ix.search(q, g["k"])
    assert_same_results(D5, I5, g["D"], g["I"], name + " pq_fast=0")
    ix.set_param("pq_fast", 1)
    ix.set_param("profile", 1)
    D6, I6 = ix.search(q, g["k"])
    assert_same_results(D6, I6, g["D"], g["I"], name + " fast")
    assert ix.get_timing("fast_queries") == g["nq"]
    auto_fallbacks = ix.get_timing("fallback_queries")
    # starve the candidate set: certificates must fail and the exact re-run must repair every query
    ix.set_param("pq_fast_kp", g["k"])
    ix.set_param("profile", 1)
    D7, I7 = ix.search(q, g["k"])
    assert_same_results(D7, I7, g["D"], g["I"], name + " starved fast scan")
    # (round 3: an uncertified query is first re-ranked from its complete candidate row — second_chance_queries — and only
    #  re-run through the exact scan if that row overflowed)
    assert ix.get_timing("fallback_queries") + ix.get_timing("second_chance_queries") > auto_fallbacks, "K' = k cannot be certifiable for every query"
    ix.set_param("pq_fast_kp", 0)
    ix.set_param("pq_filter", 0)            # fast scan through the full score buffer instead of in-kernel filtering
    D8, I8 = ix.search(q, g["k"])
    assert_same_results(D8, I8, g["D"], g["I"], name + " unfiltered fast scan")
    ix.set_param("pq_filter", 1)
    ix.set_param("pq_prepass_fused", 0)     # threshold pre-pass as grouping + k_pq_scan8 + selection launches
    Da, Ia = ix.search(q, g["k"])
    assert_same_results(Da, Ia, g["D"], g["I"], name + " multi-launch pre-pass")
    ix.set_param("pq_prepass_fused", 1)
    ix.set_param("pq_prune", 1)             # opt-in pair pruning: (list, group) items that cannot hold a survivor are skipped
    Db, Ib = ix.search(q, g["k"])
    assert_same_results(Db, Ib, g["D"], g["I"], name + " pair pruning")
    ix.set_param("pq_prune", 0)
    ix.set_param("lut_tiled", 0)            # one workgroup per query builds its table (the tiled build is the dsub = 8 default)
    D9, I9 = ix.search(q, g["k"])
    assert_same_results(D9, I9, g["D"], g["I"], name + " per-query table build")
    ix.set_param("lut_tiled", 1)


@pytest.mark.parametrize("d,M,nlist", [(96, 12, 8), (64, 8, 4), (768, 16, 16), (128, 64, 8), (64, 16, 7), (320, 160, 4),
                                       (256, 32, 8), (256, 128, 4), (192, 96, 5), (384, 64, 6), (768, -16, 16), (128, -16, 3)])
def test_ivfpq_shapes_vs_oracle(gpu, orc, d, M, nlist):
    """Other (d, M): 16-byte-granule and 4-byte-granule code layouts, dsub 8/2/48; M=160 is too large for the
    LDS-resident table build and takes the unfused table path; M = 32 / 64 / 96 / 128 take the rotated layout
    (half phase only, one full phase, full + half, two full phases of k_pq_scan_rot)."""
    gran16 = M < 0           # M = -16: the granule layout for M = 16 (on request since round 4: rsx_set_param pq_layout = 0)
    M = abs(M)
    n, nq, k = 6000, 37, 20
    x = orc.synth_vectors(d, nlist, 61, 62, 0.5, 0, n)
    q = orc.synth_queries(d, nlist, 61, 62, 0.5, n, 63, 0.1, 0, nq)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 4, 1234)
    a, _ = orc.assign_ip(cen, x32)
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:2000], M, 3, 1234)
    codes = orc.pq_encode(cb, orc.residuals(cen, x32, a))
    lm = orc.ListMajor(a, np.arange(n), codes, nlist)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    assert ix._get("pq_layout") == (0 if M not in (16, 32, 64, 96, 128) else 2 if M == 96 else 1)
    if gran16:
        ix.set_param("pq_layout", 0)
        assert ix._get("pq_layout") == 0
    ix.set_centroids(cen); ix.set_codebooks(cb)
    ix.add(x)
    for nprobe in (1, 3, nlist):
        ix.nprobe = nprobe
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, nprobe, k)
        assert_same_results(D, I, Dr, Ir, f"d={d} M={M} nprobe={nprobe}")


@pytest.mark.parametrize("layout", [2, 1, 0])
def test_ivfpq_many_survivors(gpu, orc, layout):
    """Large K' (weak threshold): tens of thousands of keys pass the in-kernel filter, survivor segments and candidate rows
    overflow — every query must then be flagged and repaired by the exact re-run, never silently lose a candidate."""
    d, n, nlist, M, nq, k = 768, 120000, 8, 96, 96, 10
    x = gpu.synth_vectors(d, 8, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 8, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.set_param("pq_layout", layout)
    ix.train(x[:20000]); ix.add(x); ix.nprobe = 4
    ix.set_param("scan_kernel", 2)
    De, Ie = ix.search(q, k)
    ix.set_param("scan_kernel", 0)
    for kp in (0, 512, 2048):
        for chunk in (0, 4096):
            ix.set_param("pq_fast_kp", kp); ix.set_param("scan_chunk", chunk)
            D, I = ix.search(q, k)
            assert_same_results(D, I, De, Ie, f"layout={layout} K'={kp} scan_chunk={chunk}")
    if layout >= 1:
        # round 4: survivors go to per-wave logs; starved logs (64 / 4 keys each) must flag every query that lost a key
        ix.set_param("pq_fast_kp", 0); ix.set_param("scan_chunk", 0)
        for cap in (64, 4):
            ix.set_param("pq_log_cap", cap); ix.set_param("profile", 1)
            D, I = ix.search(q, k)
            assert_same_results(D, I, De, Ie, f"layout={layout} pq_log_cap={cap}")
        assert ix.get_timing("fallback_overflow_queries") > 0, "4-key logs cannot hold this batch's survivors"
        ix.set_param("pq_log_cap", 0); ix.set_param("profile", 0)
    # k = 300 -> K' = 512 on its own
    ix.set_param("pq_fast_kp", 0); ix.set_param("scan_chunk", 0)
    ix.set_param("scan_kernel", 2)
    De, Ie = ix.search(q[:16], 300)
    ix.set_param("scan_kernel", 0)
    D, I = ix.search(q[:16], 300)
    assert_same_results(D, I, De, Ie, f"layout={layout} k=300")


@pytest.mark.parametrize("layout", [0, 1])
def test_ivfpq_large_k_and_ties(gpu, orc, layout):
    """layout 1: the rotated M = 16 form (64-vector blocks); two of the lists span two scan tiles of 1024 vectors."""
    d, M, nlist, n = 64, 16, 8, 5000
    x = orc.synth_vectors(d, nlist, 71, 72, 0.5, 0, n)
    x[100:140] = x[7]                    # 41 identical vectors -> identical codes -> exact score ties
    q = np.concatenate([x[7:8], orc.synth_queries(d, nlist, 71, 72, 0.5, n, 73, 0.1, 0, 9)], 0)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 4, 1234)
    a, _ = orc.assign_ip(cen, x32)
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:2000], M, 3, 1234)
    lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, orc.residuals(cen, x32, a)), nlist)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    ix.set_param("pq_layout", layout)
    ix.set_centroids(cen); ix.set_codebooks(cb); ix.add(x)
    ix.nprobe = nlist
    for k in (64, 500, 1000, 2048, 4096):
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, nlist, k)
        assert np.array_equal(D, Dr), f"k={k} scores"
        assert np.array_equal(I, Ir), f"k={k} ids (ties by id ascending)"


def test_ivfflat_skewed_lists_and_l2(gpu, orc):
    """One giant list + empty lists; fp32 storage; L2 metric through the bias path."""
    rng = np.random.RandomState(5)
    d, nlist, n = 64, 8, 5000
    cen = rng.randn(nlist, d).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    x = (cen[0] * 3 + 0.3 * rng.randn(n, d)).astype(np.float32)      # everything lands near centroid 0
    x[:50] = (cen[5] * 3 + 0.3 * rng.randn(50, d)).astype(np.float32)
    q = (x[rng.randint(0, n, 45)] + 0.05 * rng.randn(45, d)).astype(np.float32)
    a, _ = orc.assign_ip(cen, x)
    assert np.bincount(a, minlength=nlist).max() > 4000
    for metric in (0, 1):
        ix = gpu.IndexIVFFlat(None, d, nlist, metric)
        ix.set_centroids(cen)
        ix.add(x)
        assert ix.storage_dtype == "float32"
        lm = orc.ListMajor(a, np.arange(n), x, nlist)
        for nprobe in (1, 2, nlist):
            ix.nprobe = nprobe
            D, I = ix.search(q, 10)
            Dr, Ir = orc.ivfflat_search(metric, cen, lm, q, nprobe, 10)
            assert np.array_equal(I, Ir), f"metric={metric} nprobe={nprobe}"
            assert np.allclose(D, Dr, rtol=0, atol=max(1e-30, np.abs(Dr[np.isfinite(Dr)]).max() * 2 ** -22))
    # the same shape with fp16 rows: LDS-DMA list scan (k_list_scan2), plain and with in-kernel candidate filtering
    x16 = x.astype(np.float16); x16f = x16.astype(np.float32)
    a16, _ = orc.assign_ip(cen, x16f)
    lm16 = orc.ListMajor(a16, np.arange(n), x16f, nlist)
    for metric in (0, 1):
        ix = gpu.IndexIVFFlat(None, d, nlist, metric)
        ix.set_centroids(cen)
        ix.add(x16)
        assert ix.storage_dtype == "float16"
        for filt in (0, 2):
            ix.set_param("ivf_filter", filt)
            for nprobe in (2, nlist):
                ix.nprobe = nprobe
                D, I = ix.search(q, 10)
                Dr, Ir = orc.ivfflat_search(metric, cen, lm16, q, nprobe, 10)
                assert np.array_equal(I, Ir), f"fp16 rows metric={metric} nprobe={nprobe} filter={filt}"
                assert np.allclose(D, Dr, rtol=0, atol=max(1e-30, np.abs(Dr[np.isfinite(Dr)]).max() * 2 ** -22))


def test_round3_engine_knobs_never_change_a_result(gpu, orc):
    """The round-3 kernels are alternatives to older ones behind engine parameters: candidate gather + select in one launch
    (pq_gather), the four-queries-per-workgroup threshold pre-pass (pq_prepass4), 32 / 64 probing queries per IVF-Flat group
    (ivf_qtiles).  Every combination must return the bits of the exact kernels."""
    # IVF-PQ, rotated layout, a batch large enough for k_pq_prepass4 (>= 64 queries) and a ragged tail (130 = 32 x 4 + 2)
    d, n, nlist, M, nq, k = 768, 90000, 16, 96, 130, 10
    x = gpu.synth_vectors(d, 16, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 16, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:20000]); ix.add(x); ix.nprobe = 8
    ix.set_param("scan_kernel", 2)
    De, Ie = ix.search(q, k)
    ix.set_param("scan_kernel", 0)
    for gather in (1, 0):
        for pre4 in (1, 0):
            for pre_rows in (4096, 2048):
                ix.set_param("pq_gather", gather); ix.set_param("pq_prepass4", pre4); ix.set_param("pq_pre_rows", pre_rows)
                D, I = ix.search(q, k)
                assert_same_results(D, I, De, Ie, f"pq_gather={gather} pq_prepass4={pre4} pq_pre_rows={pre_rows}")
                D, I = ix.search(q[:70], 100)          # 160 k > sample rows: the one-query-per-workgroup pre-pass whatever the knob says
                ix.set_param("scan_kernel", 2); Dx, Ix = ix.search(q[:70], 100); ix.set_param("scan_kernel", 0)
                assert_same_results(D, I, Dx, Ix, f"k=100 pq_gather={gather} pq_prepass4={pre4}")
    # IVF-Flat, fp16 rows: 200 queries x nprobe 4 over 8 lists = 100 probing queries per list
    rng = np.random.RandomState(11)
    d, nlist, n, nq = 64, 8, 20000, 200
    cen = rng.randn(nlist, d).astype(np.float32)
    x = (cen[rng.randint(0, nlist, n)] + 0.4 * rng.randn(n, d)).astype(np.float16)
    qf = (x[rng.randint(0, n, nq)].astype(np.float32) + 0.05 * rng.randn(nq, d)).astype(np.float32)
    xf = x.astype(np.float32)
    a, _ = orc.assign_ip(cen, xf)
    lm = orc.ListMajor(a, np.arange(n), xf, nlist)
    for metric in (0, 1):
        ixf = gpu.IndexIVFFlat(None, d, nlist, metric)
        ixf.set_centroids(cen); ixf.add(x)
        assert ixf.storage_dtype == "float16"
        for nprobe in (4, nlist, 1):
            ixf.nprobe = nprobe
            Dr, Ir = orc.ivfflat_search(metric, cen, lm, qf, nprobe, 10)
            for qt in (1, 0, 2, 4):
                for filt in (0, 2):
                    ixf.set_param("ivf_qtiles", qt); ixf.set_param("ivf_filter", filt)
                    D, I = ixf.search(qf, 10)
                    assert np.array_equal(I, Ir), f"metric={metric} nprobe={nprobe} ivf_qtiles={qt} ivf_filter={filt}"
                    assert np.allclose(D, Dr, rtol=0, atol=max(1e-30, np.abs(Dr[np.isfinite(Dr)]).max() * 2 ** -22))


def test_ivfflat_query_stationary_scan_equals_the_oracle(gpu, orc):
    """k_list_scan3 (ivf_qtiles = 8: 128 probing queries per group held in registers, the rows of a list streamed once per group
    through a ring all eight waves read), d = 768 / 384 / 1024, inner product and squared distance (the rows' bias rides in the stages).
    Ragged groups (150 probing queries per list = 128 + 22), lists whose length is no multiple of the 128-row stage or of the 1024-row
    work item, a list shorter than one 16-row piece, filtered and score-row forms, k = 10 and k = 100: the ids of the oracle and the
    score bits of the 16-query kernel."""
    for d, n, nq, ks in ((768, 20000, 300, (10, 100)), (384, 12000, 200, (10,)), (1024, 9000, 200, (10,))):
        rng = np.random.RandomState(5 + d)
        nlist = 8
        cen = rng.randn(nlist, d).astype(np.float32)
        lab = rng.randint(0, nlist - 1, n); lab[:5] = nlist - 1                 # list 7: five rows
        x = (cen[lab] + 0.4 * rng.randn(n, d)).astype(np.float16)
        qf = (x[rng.randint(0, n, nq)].astype(np.float32) + 0.05 * rng.randn(nq, d)).astype(np.float32)
        xf = x.astype(np.float32)
        a, _ = orc.assign_ip(cen, xf)
        lm = orc.ListMajor(a, np.arange(n), xf, nlist)
        for metric in (0, 1):
            ixf = gpu.IndexIVFFlat(None, d, nlist, metric)
            ixf.set_centroids(cen); ixf.add(x)
            assert ixf.storage_dtype == "float16"
            for nprobe in (4, nlist):
                ixf.nprobe = nprobe
                for k in ks:
                    Dr, Ir = orc.ivfflat_search(metric, cen, lm, qf, nprobe, k)
                    ixf.set_param("ivf_qtiles", 0); ixf.set_param("ivf_filter", 0)
                    D1, I1 = ixf.search(qf, k)
                    assert np.array_equal(I1, Ir), f"d={d} metric={metric} nprobe={nprobe} k={k}: 16-query kernel"
                    for filt in (0, 2):
                        ixf.set_param("ivf_qtiles", 8); ixf.set_param("ivf_filter", filt)
                        D, I = ixf.search(qf, k)
                        assert np.array_equal(I, Ir), f"d={d} metric={metric} nprobe={nprobe} k={k} ivf_filter={filt}"
                        assert np.array_equal(D.view(np.uint32), D1.view(np.uint32)), f"d={d} metric={metric} nprobe={nprobe} k={k} ivf_filter={filt}: score bits"


def test_train_matches_oracle(gpu, orc):
    """rsx_train == the oracle's k-means / PQ training, bit for bit (GPU assignment, host update)."""
    d, M, nlist, n = 64, 8, 8, 3000
    x = orc.synth_vectors(d, nlist, 81, 82, 0.5, 0, n)
    x32 = x.astype(np.float32)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    ix.train(x)
    assert ix.is_trained
    cen = orc.kmeans(0, x32[orc.kmeans_sample(n, nlist, 256, 1234)], nlist, 10, 1234)
    assert np.array_equal(ix.get_centroids(), cen), "coarse centroids"
    sel = orc.kmeans_sample(n, 256, 256, 1234)
    at, _ = orc.assign_ip(cen, x32[sel])
    cb = orc.pq_train(orc.residuals(cen, x32[sel], at), M, 25, 1234)
    assert np.array_equal(ix.get_codebooks(), cb), "PQ codebooks"
    ivf = gpu.IndexIVFFlat(None, d, nlist, 0)
    ivf.train(x32)
    assert np.array_equal(ivf.get_centroids(), cen)


def test_write_read_roundtrip(gpu, orc, tmp_path):
    g = load_golden("ivfpq_d64_m16")
    x, q = regen_gpu(gpu, g)
    ix = gpu.IndexIVFPQ(None, g["d"], g["nlist"], g["M"], 8, 0)
    ix.set_centroids(g["centroids"]); ix.set_codebooks(g["codebooks"]); ix.add(x)
    p = str(tmp_path / "pq.faiss")
    gpu.write_index(ix, p)
    ix2 = gpu.read_index(p)
    assert ix2.ntotal == g["n"] and ix2.is_trained
    ix2.nprobe = g["nprobe"]
    D, I = ix2.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], "pq reload")
    gf = load_golden("flat_ip_d100")
    xf, qf = regen_gpu(gpu, gf)
    f = gpu.IndexFlatIP(gf["d"]); f.add(xf)
    p2 = str(tmp_path / "flat.faiss")
    gpu.write_index(f, p2)
    D, I = gpu.read_index(p2).search(qf, gf["k"])
    assert_same_results(D, I, gf["D"], gf["I"], "flat reload")
    gi = load_golden("ivfflat_d768")
    xi, qi = regen_gpu(gpu, gi)
    iv = gpu.IndexIVFFlat(None, gi["d"], gi["nlist"], 0)
    iv.set_centroids(gi["centroids"]); iv.add(xi)
    p3 = str(tmp_path / "ivf.faiss")
    gpu.write_index(iv, p3)
    iv2 = gpu.read_index(p3)
    iv2.nprobe = gi["nprobe"]
    D, I = iv2.search(qi, gi["k"])
    assert_same_results(D, I, gi["D"], gi["I"], "ivfflat reload")


def test_merge_kernel_matches_reference_rule(gpu, orc):
    e = load_golden("edge_cases")
    Do, Io = gpu.merge_topk(e["Dm"], e["Im"])
    assert np.array_equal(Io, e["Imo"]) and np.array_equal(Do, e["Dmo"])
    import json, os
    from util import GOLDEN
    with open(os.path.join(GOLDEN, "search_golden.json")) as f:   # output of the reference's own rerank_elements
        r = json.load(f)["rerank_elements"]
    Dg, Ig = gpu.merge_topk(np.asarray(r["D"], np.float32), np.asarray(r["I"], np.int64))
    assert Ig.tolist() == r["IDs"] and Dg.tolist() == r["scores"]
    rng = np.random.RandomState(0)
    D = np.sort(rng.randint(0, 9, size=(8, 33, 10)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    I = rng.randint(0, 10 ** 9, size=(8, 33, 10)).astype(np.int64)
    I[3, :, 6:] = -1
    Dr, Ir = orc.merge_topk(D, I, 0)
    Do, Io = gpu.merge_topk(D, I)
    assert np.array_equal(Io, Ir) and np.array_equal(Do, Dr)
    import torch
    Dt, It = gpu.merge_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda())
    assert np.array_equal(It.cpu().numpy(), Ir) and np.array_equal(Dt.cpu().numpy(), Dr)
    # the multi-GPU exchange form: per-shard pack (adds the shard's id offset, keeps -1 padding) -> [ns, 2, nq, k] -> merge
    D[2, :, 0] = -3.5                                       # a negative score: its sign bit must survive the packing
    D[2] = np.sort(D[2], axis=1)[:, ::-1]
    offs = [7 * s for s in range(8)]
    packed = torch.stack([gpu.pack_topk(torch.from_numpy(D[s]).cuda(), torch.from_numpy(I[s]).cuda(), offs[s]) for s in range(8)])
    Dp, Ip = gpu.merge_packed(packed)
    Ioff = np.stack([np.where(I[s] >= 0, I[s] + offs[s], I[s]) for s in range(8)])
    Dr2, Ir2 = orc.merge_topk(D, Ioff, 0)
    assert np.array_equal(Ip.cpu().numpy(), Ir2) and np.array_equal(Dp.cpu().numpy(), Dr2)


@pytest.mark.parametrize("kind", ["ivfpq", "ivfflat"])
def test_list_sharded_index_equals_single_index(gpu, orc, kind):
    """Three handles with add_list_mod = 3 fed the same add stream (sequential ids) + merge == one index:
    the list-sharded multi-GPU build of bench.py, emulated on one GPU."""
    d, nlist, n, nq, k = 64, 16, 9000, 50, 10
    x = orc.synth_vectors(d, nlist, 71, 72, 0.5, 0, n)
    q = orc.synth_queries(d, nlist, 71, 72, 0.5, n, 73, 0.1, 0, nq)
    x32 = x.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 3, 5)

    def make():
        if kind == "ivfpq":
            ix = gpu.IndexIVFPQ(None, d, nlist, 16, 8, 0)
            ix.set_centroids(cen); ix.set_codebooks(cb)
        else:
            ix = gpu.IndexIVFFlat(None, d, nlist, 0)
            ix.set_centroids(cen)
        ix.nprobe = 5
        return ix
    a, _ = orc.assign_ip(cen, x32)
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:2000], 16, 2, 5) if kind == "ivfpq" else None
    full = make()
    shards = [make() for _ in range(3)]
    for r, s in enumerate(shards):
        s.set_param("add_list_mod", 3); s.set_param("add_list_rem", r)
    for c0 in range(0, n, 2500):                       # several add calls: the sequential ids must keep counting dropped vectors
        full.add(x[c0:c0 + 2500])
        for s in shards:
            s.add(x[c0:c0 + 2500])
    assert sum(s.ntotal for s in shards) == full.ntotal == n
    counts = np.bincount(a, minlength=nlist)
    for r, s in enumerate(shards):
        assert s.ntotal == counts[r::3].sum()
    Df, If = full.search(q, k)
    for s in shards:
        s.set_param("profile", 1)
    Ds, Is = zip(*[s.search(q, k) for s in shards])
    if kind == "ivfpq":   # a shard whose lists do not include a query's closest list must still get a threshold (no mass fallback)
        assert sum(s.get_timing("fallback_queries") for s in shards) <= 2
    Dm, Im = gpu.merge_topk(np.stack(Ds), np.stack(Is))
    # exact cross-shard score ties may come back in shard order instead of id order; none in this data
    assert_same_results(Dm, Im, Df, If, f"list-sharded {kind}")
    if kind == "ivfpq":
        # two-call search (rsx_search_prepass / rsx_search_scan): the shards' thresholds are raised to their maximum between the
        # pre-pass and the scan — what the all-reduce(MAX) of sharded.ShardedSearcher(exchange_thresholds=True) does across
        # ranks.  Same merged bits, no exact re-runs, and no shard keeps more candidates than before.
        import torch
        from sharded import raise_thresholds
        qd = torch.from_numpy(q).cuda()
        for s in shards:
            s.set_param("profile", 2)
        plain = [s.search(qd, k) for s in shards]
        cand_plain = [s.get_timing("cand_keys") for s in shards]
        for s in shards:
            s.set_param("profile", 2)
        taus = [s.search_prepass(qd, k) for s in shards]
        assert any(t is not None for t in taus)
        raise_thresholds(taus)
        torch.cuda.synchronize()
        two = [s.search_scan() for s in shards]
        cand_two = [s.get_timing("cand_keys") for s in shards]
        assert sum(s.get_timing("fallback_queries") for s in shards) == 0
        assert all(b <= a for a, b in zip(cand_plain, cand_two)) and sum(cand_two) < sum(cand_plain)
        Dm3, Im3 = gpu.merge_topk(torch.stack([d_ for d_, _ in two]), torch.stack([i_ for _, i_ in two]))
        assert_same_results(Dm3.cpu().numpy(), Im3.cpu().numpy(), Df, If, "list-sharded ivfpq, thresholds exchanged")
        for s in shards:
            s.set_param("profile", 1)
    with pytest.raises(RuntimeError):
        shards[0].set_param("add_list_mod", 2)          # only before the first add
    # a list shard survives save / load: the vectors it dropped still count towards the sequential ids of later adds
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "shard1.faiss")
        gpu.write_index(shards[1], path)
        back = gpu.read_index(path)
    back.nprobe = 5
    extra = x[:600]
    full.add(extra); back.add(extra)
    for s in (shards[0], shards[2]):
        s.add(extra)
    Df2, If2 = full.search(q, k)
    parts = [shards[0].search(q, k), back.search(q, k), shards[2].search(q, k)]
    Dm2, Im2 = gpu.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert_same_results(Dm2, Im2, Df2, If2, f"list-sharded {kind} after reloading one shard")


def test_sharded_searcher_over_rccl_single_rank(gpu, orc):
    """The real collective path (RCCL all_gather_into_tensor on HBM tensors + merge kernel) with world_size 1;
    the 2-rank semantics are covered on CPU by tests/test_sharded_gloo.py."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from sharded import ShardedSearcher
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = load_golden("ivfpq_d64_m16")
        x, q = regen_gpu(gpu, g)
        ix = gpu.IndexIVFPQ(None, g["d"], g["nlist"], g["M"], 8, 0)
        ix.set_centroids(g["centroids"]); ix.set_codebooks(g["codebooks"]); ix.add(x)
        ix.nprobe = g["nprobe"]
        sh = ShardedSearcher(ix, id_offset=1000, force_collective=True)
        D, I = sh.search(torch.from_numpy(q).cuda(), g["k"])
        assert D.is_cuda and I.is_cuda
        assert_same_results(D.cpu().numpy(), I.cpu().numpy() - 1000, g["D"], g["I"], "sharded over RCCL")
        D2, I2 = sh.search(q, g["k"])          # host arrays in -> host arrays out
        assert_same_results(D2, I2 - 1000, g["D"], g["I"], "sharded over RCCL (numpy)")
    finally:
        dist.destroy_process_group()


def test_two_call_search_owns_the_handle(gpu, orc):
    """between rsx_search_prepass and rsx_search_scan the parked search owns the handle's workspaces.  Every other
    entry point must refuse the handle (it used to overwrite the parked search's thresholds — or park itself and hang), the
    parked search must still finish with the single-call result, and ShardedSearcher must release it when the exchange fails."""
    import torch
    d, n, nlist, M, nq, k = 96, 9000, 8, 32, 40, 10
    x = gpu.synth_vectors(d, 16, 77, 5000, 0.5, 0, n)
    q = torch.from_numpy(gpu.synth_queries(d, 16, 77, 5000, 0.5, n, 31, 0.1, 0, nq)).cuda()
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:4000]); ix.add(x); ix.nprobe = 4
    Dr, Ir = ix.search(q, k)
    tau = ix.search_prepass(q, k)
    assert tau is not None and tau.shape[0] == nq
    for call in (lambda: ix.search(q, k), lambda: ix.add(x[:10]), lambda: ix.reset(), lambda: ix.set_param("pq_filter", 0),
                 lambda: ix.train(x[:4000]), lambda: ix.search_prepass(q, k)):
        with pytest.raises(RuntimeError, match="two-call"):
            call()
    D, I = ix.search_scan()
    assert torch.equal(D, Dr) and torch.equal(I, Ir)
    D2, I2 = ix.search(q, k)            # the handle is usable again
    assert torch.equal(D2, Dr) and torch.equal(I2, Ir)
    # a batch larger than query_batch goes through the two-call form in query_batch-sized pieces; a failing exchange releases the handle
    import torch.distributed as dist
    import sharded
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        ss = sharded.ShardedSearcher(ix, exchange_thresholds=True, force_collective=True)
        ix.set_param("query_batch", 16)
        D3, I3 = ss._search_two_call(q, k)
        assert torch.equal(D3, Dr) and torch.equal(I3, Ir)
        ix.set_param("query_batch", 1024)
        real = dist.all_reduce

        def boom(*a, **kw):
            raise RuntimeError("collective timed out")
        dist.all_reduce = boom
        try:
            with pytest.raises(RuntimeError, match="timed out"):
                ss._search_two_call(q, k)
        finally:
            dist.all_reduce = real
        D4, I4 = ix.search(q, k)        # no parked worker left behind
        assert torch.equal(D4, Dr) and torch.equal(I4, Ir)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("M", [16, 96])
def test_threshold_sample_spans_short_lists(gpu, orc, M):
    """Round 4: a query whose closest list holds fewer than k vectors used to get NO filter threshold (every vector of every probed
    list survived, the candidate row overflowed, the query was re-run exactly).  The pre-pass now continues its sample in the next
    closest lists with integer sums shifted down by the coarse-score difference (a lower bound of the approximate score): same
    results, no overflow, no exact re-run."""
    d, nlist, nq, k = 768 if M == 96 else 64, 12, 24, 40
    rng = np.random.RandomState(3)
    cen = rng.randn(nlist, d).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    # lists 0..3 hold 7 / 20 / 33 / 5 vectors (fewer than k), the others thousands
    sizes = [7, 20, 33, 5] + [3000] * (nlist - 4)
    xs = [(cen[l] * 4 + 0.3 * rng.randn(sz, d)).astype(np.float16) for l, sz in enumerate(sizes)]
    x = np.concatenate(xs, 0)
    perm = rng.permutation(len(x)); x = x[perm]
    x32 = x.astype(np.float32)
    a, _ = orc.assign_ip(cen, x32)
    assert np.bincount(a, minlength=nlist)[:4].max() < k
    # queries next to the short lists' centroids (their closest list is short) and a few ordinary ones
    q = np.concatenate([(cen[l] * 4 + 0.2 * rng.randn(5, d)) for l in range(4)] + [(cen[7] * 4 + 0.2 * rng.randn(4, d))], 0).astype(np.float16)
    assert len(q) == nq
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:4000], M, 3, 1234)
    lm = orc.ListMajor(a, np.arange(len(x)), orc.pq_encode(cb, orc.residuals(cen, x32, a)), nlist)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    ix.set_centroids(cen); ix.set_codebooks(cb); ix.add(x)
    for nprobe in (3, nlist):
        ix.nprobe = nprobe
        ix.set_param("profile", 2)
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q.astype(np.float32), nprobe, k)
        assert_same_results(D, I, Dr, Ir, f"short closest lists, M={M} nprobe={nprobe}")
        assert ix.get_timing("fallback_overflow_queries") == 0
        if nprobe == nlist:     # a threshold exists: far fewer survivors than the vectors of the probed lists
            assert ix.get_timing("cand_keys_max") < 0.5 * len(x), ix.get_timing("cand_keys_max")
    ix.set_param("profile", 0)


@pytest.mark.parametrize("M,d", [(16, 768), (32, 256)])
def test_ivfpq_bulk_ties_at_the_kth_score(gpu, orc, M, d):
    """Round 4 (k_pq_final_tab): PQ ties come in bulk — vectors with identical codes in one list have bit-equal scores (at M = 16 whole
    data clusters do on the bench mixture: up to 16 000 candidates at one score) — and the order among them is id ascending.  Here
    3000 copies of one vector (scattered ids) straddle rank k: far more ties than the kernel's sort holds, so the tied candidates'
    ids go through the second radix selection.  Every k must reproduce the oracle's ids and scores; M = 16 takes the
    finalize-from-the-row path by default (dsub 48), M = 32 is forced onto it."""
    nlist, n = 8, 12000
    rng = np.random.RandomState(9)
    x = orc.synth_vectors(d, nlist, 91, 92, 0.5, 0, n)
    dup = rng.choice(n, 3000, replace=False)
    x[dup] = x[dup[0]]
    q = np.concatenate([x[dup[:1]], orc.synth_queries(d, nlist, 91, 92, 0.5, n, 93, 0.1, 0, 6)], 0)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 4, 1234)
    a, _ = orc.assign_ip(cen, x32)
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:3000], M, 3, 1234)
    lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, orc.residuals(cen, x32, a)), nlist)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    ix.set_centroids(cen); ix.set_codebooks(cb); ix.add(x)
    ix.set_param("pq_final_tab", 2)
    ix.nprobe = nlist
    for k in (10, 100, 1000, 2500, 4096):
        ix.set_param("profile", 1)
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, nlist, k)
        assert np.array_equal(D, Dr), f"M={M} k={k} scores"
        assert np.array_equal(I, Ir), f"M={M} k={k} ids (ties by id ascending)"
        assert ix.get_timing("fallback_queries") == 0, f"M={M} k={k}: settled from the candidate row, not by the exact re-run"
    ties = (Dr[0, 1:] == Dr[0, :-1]).sum()
    assert ties > 2000, "the fixture must tie in bulk"


@pytest.mark.parametrize("M,d", [(96, 768), (16, 768), (64, 256)])
def test_round4_engine_knobs_never_change_a_result(gpu, orc, M, d):
    """Round 4's alternatives behind engine parameters: one or two streams per search batch (overlap), finalize from the complete
    candidate row or through the K' cut (pq_final_tab 0 / 1 / 2), the size of the large-k threshold sample (pq_pre_mult / pq_pre_max).
    Every combination must return the bits of the exact kernel, for small and large k."""
    n, nlist, nq = 60000, 16, 130
    x = gpu.synth_vectors(d, 16, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 16, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:20000]); ix.add(x); ix.nprobe = 8
    for k in (10, 300):
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q, k)
        ix.set_param("scan_kernel", 0)
        for overlap in (1, 0):
            for tab in (1, 2, 0):
                for mult, mx in ((160, 16384), (8, 2048)):
                    ix.set_param("overlap", overlap); ix.set_param("pq_final_tab", tab)
                    ix.set_param("pq_pre_mult", mult); ix.set_param("pq_pre_max", mx)
                    D, I = ix.search(q, k)
                    assert_same_results(D, I, De, Ie, f"M={M} k={k} overlap={overlap} pq_final_tab={tab} sample={mult}x/{mx}")


@pytest.mark.parametrize("M,d,nlist", [(96, 768, 16), (16, 768, 40), (64, 256, 7), (20, 160, 16), (6, 48, 9), (128, 1024, 12)])
def test_round6_table_build_and_grouping_forms_never_change_a_result(gpu, orc, M, d, nlist):
    """Round 6: the 8-bit tables from the matrix cores (lut_tiled 2: k_pq_lut_mfma, per-query parameters in its tail) against the VALU
    forms (1: two passes + k_pq_qparam, 0: one workgroup per query), the pair grouping as extra workgroups of the table launch
    (pq_group_fused) against its own launches, one stream against two — every combination returns the exact kernel's bits; batch sizes
    around the 32-query tile of the matrix-core form, several searches in a row (the hand-over words and tile counters carry over)."""
    n = 50000
    x = gpu.synth_vectors(d, 16, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 16, 1234, 10000, 0.5, n, 999, 0.1, 0, 200)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:20000]); ix.add(x); ix.nprobe = min(8, nlist)
    for nq, k in ((200, 10), (33, 10), (1, 5), (97, 120), (64, 10)):
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q[:nq], k)
        ix.set_param("scan_kernel", 0)
        for overlap, tiled, fused in ((0, 2, 1), (0, 2, 0), (1, 2, 1), (0, 1, 1), (1, 1, 0), (0, 0, 1), (0, 2, 1)):
            ix.set_param("overlap", overlap); ix.set_param("lut_tiled", tiled); ix.set_param("pq_group_fused", fused)
            ix.set_param("pq_lut_early", (overlap + tiled + fused) & 1)        # the table build's first pass in the probe-pick launch or on its own
            D, I = ix.search(q[:nq], k)
            assert_same_results(D, I, De, Ie, f"M={M} nq={nq} k={k} overlap={overlap} lut_tiled={tiled} pq_group_fused={fused}")
    ix.set_param("overlap", 0); ix.set_param("lut_tiled", 2); ix.set_param("pq_group_fused", 1); ix.set_param("pq_lut_early", 1)


@pytest.mark.parametrize("kind", ["ivfpq", "ivfflat"])
@pytest.mark.parametrize("d,nlist,nprobe", [(768, 256, 32), (96, 130, 7), (100, 64, 32), (64, 1024, 48), (128, 4500, 20), (64, 300, 49)])
def test_round6_fast_coarse_quantiser_is_exact(gpu, orc, kind, d, nlist, nprobe):
    """Round 6: the coarse quantiser from fp16 MFMA scores + exact chains of the candidates (k_coarse_pick) must select exactly the lists —
    and pass on exactly the coarse scores — of the exact GEMM: results equal the oracle's and the coarse_fast = 0 run's bit for bit, for
    fp16 and fp32 queries, near-duplicate centroids (ties within the rounding of the approximate scores), an exact duplicate (tie by list
    number), a query of huge norm, one with a NaN, and — where the index has the lists for it — a cluster of 80 nearly equal centroids that
    overflows the candidate rows of the queries near it: those leave through the exact re-run (counted), the batch must not.  nprobe 49 is
    one more than the fast form serves: the exact GEMM takes it."""
    rng = np.random.RandomState(5)
    n, nq, k = 40000, 70, 10
    x = gpu.synth_vectors(d, 64, 1234, 10000, 0.5, 0, n)
    x32 = x.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 2, 7)
    cen[1::7] = cen[0::7][: len(cen[1::7])] * (1.0 + 2e-4 * rng.randn(len(cen[1::7]), 1).astype(np.float32))     # near-duplicates of other centroids
    cen[5] = cen[4]                                                                                               # an exact duplicate: tie by list number
    crowd = nlist >= 256 and nprobe <= 100
    if crowd:
        cen[100:180] = cen[99] * (1.0 + 1e-5 * rng.randn(80, 1).astype(np.float32))
    q = gpu.synth_queries(d, 64, 1234, 10000, 0.5, n, 999, 0.1, 0, nq).astype(np.float32)
    q[9] *= 3000.0
    if crowd:
        q[20:23] = cen[99] * (1.0 + 0.01 * rng.randn(3, d).astype(np.float32))
    q32 = q + (1e-3 * rng.randn(nq, d)).astype(np.float32)      # not fp16-representable
    a, _ = orc.assign_ip(cen, x32)
    if kind == "ivfpq":
        M = 16 if d % 16 == 0 else 4
        ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
        ix.set_centroids(cen)
        res = orc.residuals(cen, x32, a)
        cb = orc.pq_train(res[:3000], M, 1, 7)
        ix.set_codebooks(cb)
        lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, res), nlist)
    else:
        ix = gpu.IndexIVFFlat(None, d, nlist, gpu.METRIC_INNER_PRODUCT)
        ix.set_centroids(cen)
        lm = orc.ListMajor(a, np.arange(n), x32, nlist)
    ix.add(x); ix.nprobe = nprobe
    for qq, what in ((q.astype(np.float16), "fp16 queries"), (q32, "fp32 queries")):
        qf = qq.astype(np.float32)
        if kind == "ivfpq":
            Dr, Ir = orc.ivfpq_search(cen, cb, lm, qf, nprobe, k)
        else:
            Dr, Ir = orc.ivfflat_search(0, cen, lm, qf, nprobe, k)
        ix.set_param("profile", 1)
        ix.set_param("coarse_fast", 1)
        D1, I1 = ix.search(qq, k)
        redo = ix.get_timing("coarse_redo_queries")
        ix.set_param("coarse_fast", 0)
        D0, I0 = ix.search(qq, k)
        ix.set_param("coarse_fast", 1); ix.set_param("profile", 0)
        assert_same_results(D0, I0, Dr, Ir, f"{kind} {what}: exact coarse quantiser vs oracle")
        assert_same_results(D1, I1, Dr, Ir, f"{kind} {what}: fast coarse quantiser vs oracle")
        assert (3 if crowd and nprobe <= 48 else 0) <= redo <= 20, f"{kind} {what}: {redo} queries re-run for the coarse quantiser"
    qn = q.copy(); qn[8, 0] = np.nan
    ix.set_param("coarse_fast", 1); D1, I1 = ix.search(qn, k)
    ix.set_param("coarse_fast", 0); D0, I0 = ix.search(qn, k)
    ix.set_param("coarse_fast", 1)
    assert np.array_equal(I1, I0) and np.array_equal(D1, D0, equal_nan=True), f"{kind}: a query with a NaN"
    # fp32 queries whose components sit below fp16's normal range (2^-14): their fp16 copies carry absolute, not relative, rounding errors —
    # the bound has a term for that (such queries mostly leave through the exact re-run; the result must be the oracle's either way)
    qt = (q32 * np.float32(2e-5)).astype(np.float32)
    if kind == "ivfpq":
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, qt, nprobe, k)
    else:
        Dr, Ir = orc.ivfflat_search(0, cen, lm, qt, nprobe, k)
    D1, I1 = ix.search(qt, k)
    assert_same_results(D1, I1, Dr, Ir, f"{kind}: tiny fp32 queries, fast coarse quantiser vs oracle")


@pytest.mark.parametrize("M", [96, 16])
def test_one_call_of_many_batches_equals_batch_by_batch(gpu, M):
    """One index.search call with several internal batches (the reference hands ALL its queries to one call, src/search.py:296;
    api_search.hip: search_impl walks them in query_batch pieces).  The results must be those of one call per batch — CUDA-tensor and
    host queries, ragged last batch — also after the index grew between two calls (the lists are re-laid out), and the exact
    kernel's."""
    import torch
    d, n, nlist, nq = 768, 80000, 32, 64 * 7 + 19
    x = gpu.synth_vectors(d, 64, 1234, 10000, 0.5, 0, n + 40000)
    q = gpu.synth_queries(d, 64, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:20000]); ix.add(x[:n]); ix.nprobe = 8
    qd = torch.from_numpy(q).cuda()
    for round_ in range(2):
        for k in (10, 200):
            ix.set_param("query_batch", 1024)
            parts = [ix.search(q[b:b + 64], k) for b in range(0, nq, 64)]
            Ds, Is = np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts])
            ix.set_param("query_batch", 64)
            D, I = ix.search(q, k)
            assert_same_results(D, I, Ds, Is, f"M={M} k={k} host queries, round {round_}")
            Dd, Id = ix.search(qd, k)
            assert_same_results(Dd.cpu().numpy(), Id.cpu().numpy(), Ds, Is, f"M={M} k={k} device queries, round {round_}")
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q, 10)
        ix.set_param("scan_kernel", 0)
        D, I = ix.search(q, 10)
        assert_same_results(D, I, De, Ie, f"M={M} internal batches vs the exact kernel, round {round_}")
        ix.add(x[n:])                              # lists overflow: the payload moves


@pytest.mark.parametrize("M,d", [(96, 768), (-96, 768), (32, 256), (64, 256), (128, 256)])
def test_large_k_prepass_histogram_form(gpu, M, d):
    """Large k on the block layouts (M = 96: sliced, the default; -96: rotated): the threshold pre-pass in its four-queries-per-workgroup histogram form
    (k_pq_prepass4<.., BIG>: samples of up to 32768 rows spanning up to eight lists, the threshold at the lower edge of the
    histogram bin that holds the k-th largest integer sum) against the one-query form (pq_prepass4 = 2), no pre-pass kernel of
    this family (0) and the exact kernel.  Lists from a few rows to tens of thousands: closest lists shorter than k, than the
    sample, longer than it.  No query may need the exact re-run."""
    rng = np.random.RandomState(3)
    nlist, n, nq = 40, 200_000, 133
    cen = rng.randn(nlist, d).astype(np.float32)
    pl = rng.dirichlet(np.full(nlist, 0.3))                 # very uneven lists
    x = (cen[rng.choice(nlist, n, p=pl)] + 0.6 * rng.randn(n, d)).astype(np.float16)
    q = (x[rng.randint(0, n, nq)].astype(np.float32) + 0.2 * rng.randn(nq, d)).astype(np.float16)
    rotated96 = M < 0
    M = abs(M)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    assert ix._get("pq_layout") == (2 if M == 96 else 1)
    if rotated96:
        ix.set_param("pq_layout", 1)
    ix.train(x[:40000]); ix.add(x); ix.nprobe = 16
    ls = ix.list_sizes()
    assert ls.min() < 1000 and ls.max() > 5000, "the test wants lists shorter than k and lists longer than the sample"
    for k in (100, 1000, 2000):
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q, k)
        ix.set_param("scan_kernel", 0)
        for pre4 in (1, 2, 0):
            for mult, mx in ((160, 16384), (160, 32768), (4, 1024)):
                ix.set_param("pq_prepass4", pre4); ix.set_param("pq_pre_mult", mult); ix.set_param("pq_pre_max", mx); ix.set_param("profile", 1)
                D, I = ix.search(q, k)
                assert_same_results(D, I, De, Ie, f"M={M} k={k} pq_prepass4={pre4} sample={mult}x/{mx}")
                assert ix.get_timing("fallback_queries") == 0, f"M={M} k={k} pq_prepass4={pre4} sample={mult}x/{mx}"
        ix.set_param("profile", 0); ix.set_param("pq_prepass4", 1); ix.set_param("pq_pre_mult", 160); ix.set_param("pq_pre_max", 16384)


def test_m64_ragged_groups_tiles_and_starved_logs(gpu):
    """The filtered M = 64 scan (k_pq_scan_rot<1, 0, true>: ONE 64 KiB table plane).  Ragged groups (a list probed by 1 .. 70
    queries), lists of several tiles (scan_chunk), small and large k, starved survivor logs (overflow -> exact re-run): always the
    exact kernel's ids and scores.  (Round 4's eight-queries-per-pass variant of this kernel was measured 7-10 % slower and
    removed in round 5: profiles/r04_rot8_m64.md.)"""
    d, M, n, nlist, nq = 256, 64, 120_000, 24, 203
    x = gpu.synth_vectors(d, 24, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 24, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    assert ix._get("pq_layout") == 1
    ix.train(x[:30000]); ix.add(x); ix.nprobe = 8
    for k in (10, 200, 1000):
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q, k)
        ix.set_param("scan_kernel", 0)
        for chunk, logcap in ((0, 0), (2048, 0), (0, 64)):
            ix.set_param("scan_chunk", chunk); ix.set_param("pq_log_cap", logcap); ix.set_param("profile", 1)
            D4, I4 = ix.search(q, k)
            fb = ix.get_timing("fallback_queries")
            ix.set_param("profile", 0)
            assert_same_results(D4, I4, De, Ie, f"k={k} scan_chunk={chunk} pq_log_cap={logcap}: fast scan vs the exact kernel")
            if logcap == 0:
                assert fb == 0, f"k={k} scan_chunk={chunk}: no exact re-run expected"
        D1, I1 = ix.search(q[:5], k)            # a handful of queries: mostly one-query records
        assert_same_results(D1, I1, De[:5], Ie[:5], f"k={k}: five queries")
    ix.set_param("scan_chunk", 0); ix.set_param("pq_log_cap", 0)


def test_ivfpq_eight_query_gathers_equal_the_exact_scan(gpu, orc):
    """Round 6: M = 64 scans EIGHT queries per table gather (8-byte entries, ds_read_b64; work items of two 4-query records).
    Ragged groups (1 .. 8 queries per list group), several tiles per list, both candidate paths (gather + select in one launch,
    compaction), starved survivor logs: always the bits of the exact scan and of the 4-query form."""
    d, n, nlist, M = 512, 70000, 12, 64
    x = gpu.synth_vectors(d, 12, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 12, 1234, 10000, 0.5, n, 999, 0.1, 0, 203)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.train(x[:20000]); ix.add(x)
    for nq, nprobe, k in ((203, 5, 10), (77, 12, 10), (9, 3, 10), (130, 4, 100)):
        ix.nprobe = nprobe
        ix.set_param("scan_kernel", 2)
        De, Ie = ix.search(q[:nq], k)
        ix.set_param("scan_kernel", 0)
        fb = {}
        for q8 in (1, 0):
            for gather in (1, 0):
                for chunk in (0, 2048):
                    ix.set_param("pq_q8", q8); ix.set_param("pq_gather", gather); ix.set_param("scan_chunk", chunk)
                    ix.set_param("profile", 1)
                    D, I = ix.search(q[:nq], k)
                    assert_same_results(D, I, De, Ie, f"nq={nq} nprobe={nprobe} k={k} pq_q8={q8} pq_gather={gather} scan_chunk={chunk}")
                    fb[(q8, gather, chunk)] = (ix.get_timing("fallback_queries"), ix.get_timing("second_chance_queries"))
        # a scan that lost survivors would be repaired by the exact re-run and still pass above: both forms must need the same repairs
        for (q8, gather, chunk), v in fb.items():
            assert v == fb[(0, gather, chunk)], f"nq={nq} nprobe={nprobe} k={k}: repairs {fb}"
        ix.set_param("pq_q8", 1); ix.set_param("pq_gather", 1); ix.set_param("scan_chunk", 0)
    ix.nprobe = 5
    ix.set_param("scan_kernel", 2)
    De, Ie = ix.search(q, 10)
    ix.set_param("scan_kernel", 0)
    for cap in (64, 4):
        ix.set_param("pq_log_cap", cap)
        D, I = ix.search(q, 10)
        assert_same_results(D, I, De, Ie, f"pq_q8 pq_log_cap={cap}")
    ix.set_param("pq_log_cap", 0); ix.set_param("profile", 0)


@pytest.mark.parametrize("nlist,n", [(6, 100000), (40, 60000)])
def test_ivfpq_sliced_layout_eight_query_scan(gpu, orc, nlist, n):
    """Round 6, M = 96: the sliced code layout (32-vector blocks cut into 32-sub-quantiser slices) and its scan k_pq_scan_sl8 —
    eight queries per table gather, two table slots in LDS, the third slice re-staged once per sub-tile, partial sums parked in
    registers.  Long lists (several sub-tiles and, with a small scan chunk, several tiles per list), short lists (a ragged last
    sub-tile), ragged query groups, both candidate paths, both threshold pre-passes, starved survivor logs, large k: always the
    bits of the exact scan, the same repairs as the rotated layout's 4-query scan, and the rotated layout's results."""
    d, M = 768, 96
    x = gpu.synth_vectors(d, nlist, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, nlist, 1234, 10000, 0.5, n, 999, 0.1, 0, 203)
    res = {}
    for layout in (2, 1):
        ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
        ix.set_param("pq_layout", layout)
        assert ix._get("pq_layout") == layout
        ix.train(x[:20000]); ix.add(x)
        for nq, nprobe, k in ((203, 5, 10), (77, min(nlist, 12), 10), (9, 3, 10), (130, 4, 100), (66, 3, 1000), (5, 1, 10)):
            ix.nprobe = nprobe
            ix.set_param("scan_kernel", 2)
            De, Ie = ix.search(q[:nq], k)
            ix.set_param("scan_kernel", 0)
            # sliced layout: the eight-query multi-pass scan (pq_q8 = 2: k_pq_scan_sl8) AND the four-query single-pass one (0: k_pq_scan_sl4);
            # the library chooses between them by batch and list size (1)
            for q8 in ((2, 0, 1) if layout == 2 else (1,)):
                for gather in (1, 0):
                    for chunk in (0, 4096):
                        for pre4 in (1, 0):
                            ix.set_param("pq_q8", q8); ix.set_param("pq_gather", gather); ix.set_param("scan_chunk", chunk); ix.set_param("pq_prepass4", pre4)
                            ix.set_param("profile", 1)
                            D, I = ix.search(q[:nq], k)
                            tag = f"layout={layout} nq={nq} nprobe={nprobe} k={k} pq_q8={q8} pq_gather={gather} scan_chunk={chunk} pq_prepass4={pre4}"
                            assert_same_results(D, I, De, Ie, tag)
                            key = (nq, nprobe, k, gather, chunk, pre4)
                            rep = (ix.get_timing("fallback_queries"), ix.get_timing("second_chance_queries"))
                            if layout == 2 and q8 == 2:
                                res[key] = (D, I, rep)
                            else:   # a scan that lost survivors would be repaired by the exact re-run: every form must need the same repairs
                                assert np.array_equal(res[key][0], D) and np.array_equal(res[key][1], I), tag
                                assert res[key][2] == rep, f"{tag}: repairs {res[key][2]} (sliced, eight queries) vs {rep}"
            ix.set_param("pq_q8", 1); ix.set_param("pq_gather", 1); ix.set_param("scan_chunk", 0); ix.set_param("pq_prepass4", 1)
        if layout == 2:
            ix.nprobe = 5
            ix.set_param("scan_kernel", 2)
            De, Ie = ix.search(q, 10)
            ix.set_param("scan_kernel", 0)
            for q8 in (2, 0):
                for cap in (64, 4):
                    ix.set_param("pq_q8", q8); ix.set_param("pq_log_cap", cap)
                    D, I = ix.search(q, 10)
                    assert_same_results(D, I, De, Ie, f"sliced pq_q8={q8} pq_log_cap={cap}")
            ix.set_param("pq_log_cap", 0); ix.set_param("pq_q8", 1)
            # codes come back in list order through the layout-independent export
            cen, cb = ix.get_centroids(), ix.get_codebooks()
            x32 = x.astype(np.float32)
            a, _ = orc.assign_ip(cen, x32[:3000])
            codes = orc.pq_encode(cb, orc.residuals(cen, x32[:3000], a))
            for l in range(min(nlist, 4)):
                c, ids = ix.get_list(l)
                sel = ids < 3000
                assert np.array_equal(c[sel], codes[ids[sel]]), f"list {l} codes through the sliced layout"


@pytest.mark.parametrize("d,M,nlist,layout", [(768, 96, 16, 2), (768, 96, 16, 1), (768, 96, 16, 0), (128, 64, 8, 1), (64, 16, 7, 1), (96, 12, 8, 0), (256, 32, 5, 1)])
def test_ivfpq_l2_metric_vs_oracle(gpu, orc, d, M, nlist, layout):
    """IndexIVFPQ(IndexFlatIP, ..., METRIC_L2) — the metric the reference never passes (src/indicies/ivf_pq.py:147-153 builds METRIC_INNER_PRODUCT)
    but `north_star` names: squared distance to the decoded vector c_l + r^, lists probed by the inner-product quantiser, results by distance
    ascending with ties by id.  Every code layout, ragged lists, duplicated vectors (exact distance ties), k beyond the probed vectors (padding
    -1 / +inf), nprobe = nlist equal to the exhaustive decode-and-compare; ids and fp32 distances bit-equal to orc_ivfpq_search_l2."""
    n, nq = 5000, 33
    x = orc.synth_vectors(d, nlist, 81, 82, 0.5, 0, n)
    x[200:230] = x[9]                                  # identical vectors -> identical codes -> exact ties
    q = np.concatenate([x[9:10], orc.synth_queries(d, nlist, 81, 82, 0.5, n, 83, 0.1, 0, nq - 1)], 0)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 4, 1234)
    a, _ = orc.assign_ip(cen, x32)
    cb = orc.pq_train(orc.residuals(cen, x32, a)[:2000], M, 3, 1234)
    codes = orc.pq_encode(cb, orc.residuals(cen, x32, a))
    lm = orc.ListMajor(a, np.arange(n), codes, nlist)
    ix = gpu.IndexIVFPQ(gpu.IndexFlatIP(d), d, nlist, M, 8, gpu.METRIC_L2)
    if ix._get("pq_layout") != layout:
        ix.set_param("pq_layout", layout)
    assert ix._get("pq_layout") == layout and ix.metric_type == gpu.METRIC_L2
    ix.set_centroids(cen); ix.set_codebooks(cb)
    ix.add(x[:1234]); ix.add(x[1234:])
    for l in range(nlist):                            # same lists and codes as the inner-product index: the metric only changes the search
        c, ids = ix.get_list(l)
        assert np.array_equal(ids, np.nonzero(a == l)[0]) and np.array_equal(c, codes[a == l])
    for nprobe, k in ((1, 10), (3, 20), (nlist, 50), (2, 4000)):
        ix.nprobe = nprobe
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, nprobe, k, metric=1)
        assert_same_results(D, I, Dr, Ir, f"IVF-PQ L2 d={d} M={M} layout={layout} nprobe={nprobe} k={k}")
    # nprobe = nlist: the exhaustive answer — distances to the decoded vectors, checked against numpy in fp64
    ix.nprobe = nlist
    D, I = ix.search(q[:5], 10)
    dsub = d // M
    dec = cen[a] + cb[np.arange(M)[None, :], codes].reshape(n, d)
    dist = ((q32[:5, None, :].astype(np.float64) - dec[None].astype(np.float64)) ** 2).sum(-1)
    for qi in range(5):
        want = np.sort(dist[qi])[:10]
        assert np.allclose(D[qi], want, rtol=2e-5, atol=1e-4), "distances are those to the decoded vectors"
    # save / load keeps the metric
    import tempfile, os
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "l2.rsx")
        gpu.write_index(ix, path)
        jx = gpu.read_index(path)
        assert jx.metric_type == gpu.METRIC_L2
        jx.nprobe = 3
        D2, I2 = jx.search(q, 20)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q32, 3, 20, metric=1)
        assert_same_results(D2, I2, Dr, Ir, "IVF-PQ L2 after save / load")


@pytest.mark.parametrize("layout", [2, 1])
def test_large_k_finalize_from_the_row_major_code_copy(gpu, orc, layout):
    """K' >= 256 (the reference's n_docs = 100 ... 2000): the finalize kernels fetch a candidate's code bytes from a row-major copy of the
    codes that is built on demand and dropped by the next add (rsx_set_param "pq_plain_codes").  Same bits with and without it, against
    the exact kernel, across adds (the copy must follow the index), list growth (re-layout) and save / load."""
    d, n, nlist, M = 768, 60000, 10, 96
    x = gpu.synth_vectors(d, 10, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, 10, 1234, 10000, 0.5, n, 999, 0.1, 0, 70)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
    ix.set_param("pq_layout", layout)
    ix.train(x[:20000]); ix.add(x[:25000]); ix.nprobe = 4
    for rnd, upto in enumerate((25000, 40000, n)):
        if rnd:
            ix.add(x[(25000, 40000)[rnd - 1]:upto])           # the copy of the previous state is stale now
        for k in (100, 1000, 2000):
            ix.set_param("scan_kernel", 2)
            De, Ie = ix.search(q, k)
            ix.set_param("scan_kernel", 0)
            for plain in (1, 0, 1):
                ix.set_param("pq_plain_codes", plain); ix.set_param("profile", 1)
                D, I = ix.search(q, k)
                assert_same_results(D, I, De, Ie, f"layout={layout} ntotal={upto} k={k} pq_plain_codes={plain}")
        ix.set_param("profile", 1); ix.search(q, 100)
        assert ix.get_timing("plain_codes_builds") <= 1, "the copy is built once per index state, not per search"
    import os, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "i.rsx")
        gpu.write_index(ix, path)
        jx = gpu.read_index(path); jx.nprobe = 4
        D, I = jx.search(q, 1000)
        ix.set_param("scan_kernel", 2); De, Ie = ix.search(q, 1000); ix.set_param("scan_kernel", 0)
        assert_same_results(D, I, De, Ie, f"layout={layout} after save / load, k = 1000")
