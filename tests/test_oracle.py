"""CPU: the oracle against the numpy fp64 brute force, its own invariants, and the golden fixtures."""
import numpy as np
import pytest

from util import load_golden, regen


@pytest.mark.parametrize("name", ["flat_ip_d768", "flat_l2_d64", "flat_ip_d100"])
def test_golden_flat(orc, name):
    g = load_golden(name)
    x, q = regen(orc, g)
    D, I = orc.flat_search(q.astype(np.float32), x.astype(np.float32), g["k"], g["metric"])
    assert np.array_equal(I, g["I"]) and np.array_equal(D, g["D"])
    D0, I0 = orc.np_flat_search(q, x, g["k"], "ip" if g["metric"] == 0 else "l2")
    assert np.array_equal(I0, g["I"]) and np.array_equal(D0, g["D"])


def test_golden_ivfflat(orc):
    g = load_golden("ivfflat_d768")
    x, q = regen(orc, g)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    a, _ = orc.assign_ip(g["centroids"], x32)
    lm = orc.ListMajor(a, np.arange(g["n"]), x32, g["nlist"])
    D, I = orc.ivfflat_search(0, g["centroids"], lm, q32, g["nprobe"], g["k"])
    assert np.array_equal(I, g["I"]) and np.array_equal(D, g["D"])


@pytest.mark.parametrize("name", ["ivfpq_d64_m16", "ivfpq_d768_m96"])
def test_golden_ivfpq(orc, name):
    g = load_golden(name)
    x, q = regen(orc, g)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    a, _ = orc.assign_ip(g["centroids"], x32)
    codes = orc.pq_encode(g["codebooks"], orc.residuals(g["centroids"], x32, a))
    from util import sha
    assert sha(a) == g["assign_sha"] and sha(codes) == g["codes_sha"]
    lm = orc.ListMajor(a, np.arange(g["n"]), codes, g["nlist"])
    D, I = orc.ivfpq_search(g["centroids"], g["codebooks"], lm, q32, g["nprobe"], g["k"])
    assert np.array_equal(I, g["I"]) and np.array_equal(D, g["D"])
    # FAISS-structured heap variant: same scores; same ids wherever scores are distinct
    Dh, Ih = orc.ivfpq_search(g["centroids"], g["codebooks"], lm, q32, g["nprobe"], g["k"], heap=True)
    assert np.array_equal(Dh, D)
    distinct = np.ones_like(D, dtype=bool)
    distinct[:, 1:] &= D[:, 1:] != D[:, :-1]
    distinct[:, :-1] &= D[:, :-1] != D[:, 1:]
    assert np.array_equal(Ih[distinct], I[distinct])


def test_adc_matches_decoded_inner_product(orc):
    """score = <q,c> + sum_m T[m][code_m] must equal <q, c + decode(code)> up to fp32 rounding."""
    g = load_golden("ivfpq_d64_m16")
    x, q = regen(orc, g)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen, cb = g["centroids"], g["codebooks"]
    a, _ = orc.assign_ip(cen, x32)
    codes = orc.pq_encode(cb, orc.residuals(cen, x32, a))
    M, dsub = cb.shape[0], cb.shape[2]
    dec = np.concatenate([cb[m, codes[:, m]] for m in range(M)], axis=1) + cen[a]
    T = orc.pq_lut(cb, q32)
    i = np.arange(0, g["n"], 97)
    for qi in range(4):
        adc = (q32[qi] @ cen[a[i]].T) + sum(T[qi, m, codes[i, m]] for m in range(M))
        ref = dec[i].astype(np.float64) @ q32[qi].astype(np.float64)
        assert np.allclose(adc, ref, rtol=1e-4, atol=1e-3)


def test_ivfpq_l2_is_the_distance_to_the_decoded_vector(orc):
    """orc_ivfpq_search_l2 (IndexIVFPQ over the inner-product quantiser with METRIC_L2): with every list probed the result is the
    brute-force L2 ranking of the DECODED vectors c_l + r^ (numpy fp64), distances to fp32 rounding; with nprobe < nlist the same
    ranking restricted to the lists the inner-product quantiser picks; ties by id; padding -1 / +inf."""
    g = load_golden("ivfpq_d64_m16")
    x, q = regen(orc, g)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    cen, cb = g["centroids"], g["codebooks"]
    a, _ = orc.assign_ip(cen, x32)
    codes = orc.pq_encode(cb, orc.residuals(cen, x32, a))
    M = cb.shape[0]
    dec = (np.concatenate([cb[m, codes[:, m]] for m in range(M)], axis=1) + cen[a]).astype(np.float64)
    lm = orc.ListMajor(a, np.arange(g["n"]), codes, g["nlist"])
    k = 10
    D, I = orc.ivfpq_search(cen, cb, lm, q32, g["nlist"], k, metric=1)
    dist = ((q32[:, None, :].astype(np.float64) - dec[None]) ** 2).sum(-1)
    for qi in range(q32.shape[0]):
        want = np.sort(dist[qi])[:k]
        assert np.allclose(D[qi], want, rtol=1e-5, atol=1e-4)
        assert np.all(np.diff(D[qi]) >= 0)
        assert np.allclose(dist[qi, I[qi]], D[qi], rtol=1e-5, atol=1e-4)
    npb = 3
    pid, _ = orc.coarse_probe(cen, q32, npb)
    D3, I3 = orc.ivfpq_search(cen, cb, lm, q32, npb, k, metric=1)
    for qi in range(q32.shape[0]):
        ok = np.isin(a, pid[qi])
        cand = np.nonzero(ok)[0]
        order = cand[np.lexsort((cand, dist[qi, cand]))][:k]
        got = I3[qi][I3[qi] >= 0]
        assert np.allclose(dist[qi, got], D3[qi][: len(got)], rtol=1e-5, atol=1e-4)
        assert set(got.tolist()) == set(order[: len(got)].tolist()) or np.allclose(np.sort(dist[qi, got]), np.sort(dist[qi, order[: len(got)]]), rtol=1e-6)
    Dp, Ip = orc.ivfpq_search(cen, cb, lm, q32[:2], 1, 5000, metric=1)      # k beyond the probed vectors
    assert (Ip[:, -1] == -1).all() and np.isposinf(Dp[:, -1]).all()


def test_nprobe_all_lists_is_exhaustive(orc):
    g = load_golden("ivfflat_d768")
    x, q = regen(orc, g)
    x32, q32 = x.astype(np.float32), q.astype(np.float32)
    a, _ = orc.assign_ip(g["centroids"], x32)
    lm = orc.ListMajor(a, np.arange(g["n"]), x32, g["nlist"])
    D, I = orc.ivfflat_search(0, g["centroids"], lm, q32, g["nlist"], g["k"])
    D0, I0 = orc.flat_search(q32, x32, g["k"], 0)
    assert np.array_equal(I, I0) and np.array_equal(D, D0)


def test_edge_cases(orc):
    e = load_golden("edge_cases")
    x, q = e["x"].astype(np.float32), e["q"].astype(np.float32)
    D, I = orc.flat_search(q, x, 8, 0)
    assert np.array_equal(I, e["I"]) and np.array_equal(D, e["D"])
    assert I[0, :4].tolist() == [3, 10, 25, 39]            # duplicates tie -> id ascending
    Dk, Ik = orc.flat_search(q, x[:5], 8, 0)
    assert (Ik[:, 5:] == -1).all() and np.isneginf(Dk[:, 5:]).all()   # k > ntotal padding
    a, _ = orc.assign_ip(e["cen"], x)
    assert np.array_equal(a, e["assign"])
    lm = orc.ListMajor(a, np.arange(40), x, 8)
    assert (np.diff(lm.list_off) == 0).any()                # some lists are empty
    Div, Iiv = orc.ivfflat_search(0, e["cen"], lm, q, 8, 8)
    assert np.array_equal(Iiv, e["Iiv"]) and np.array_equal(Iiv, I)


def test_merge_semantics(orc):
    e = load_golden("edge_cases")
    Do, Io = orc.merge_topk(e["Dm"], e["Im"], 0)
    assert np.array_equal(Io, e["Imo"]) and np.array_equal(Do, e["Dmo"])
    assert Io[0].tolist() == [10, 20, 30]   # 5.0 (shard0), 5.0 (shard1), 4.0 (shard2): earlier shard first
    # the product's host restatement of the same rule
    from sharded import merge_topk_host
    Dh, Ih = merge_topk_host(e["Dm"], e["Im"], 0)
    assert np.array_equal(Ih, Io) and np.array_equal(Dh, Do)
    rng = np.random.RandomState(0)
    D = np.sort(rng.randint(0, 6, size=(4, 9, 5)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    I = rng.randint(0, 1000, size=(4, 9, 5)).astype(np.int64)
    I[2, :, 3:] = -1
    Do, Io = orc.merge_topk(D, I, 0)
    Dh, Ih = merge_topk_host(D, I, 0)
    assert np.array_equal(Ih, Io) and np.array_equal(Dh, Do)


def test_synth_is_deterministic_and_chunkable(orc):
    a = orc.synth_vectors(48, 7, 1, 2, 0.5, 0, 300)
    b = np.concatenate([orc.synth_vectors(48, 7, 1, 2, 0.5, 0, 100), orc.synth_vectors(48, 7, 1, 2, 0.5, 100, 200)])
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    f = a.astype(np.float32)
    assert abs(f.mean()) < 0.1 and 0.8 < f.std() < 1.4


def test_kmeans_properties(orc):
    x = orc.synth_vectors(32, 8, 5, 6, 0.3, 0, 2000).astype(np.float32)
    c = orc.kmeans(0, x, 8, 10, 1234)
    assert np.allclose(np.linalg.norm(c, axis=1), 1.0, atol=1e-5)      # spherical (IP quantiser)
    c2 = orc.kmeans(0, x, 8, 10, 1234)
    assert np.array_equal(c, c2)
    cl = orc.kmeans(1, x, 8, 10, 1234)
    a = np.argmin(((x[:, None, :] - cl[None]) ** 2).sum(-1), axis=1)
    assert len(np.unique(a)) == 8
