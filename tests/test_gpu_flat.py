"""GPU: Flat index (IndexFlatIP / L2) through the C-ABI against the oracle and the golden fixtures."""
import numpy as np
import pytest

from util import assert_same_results, load_golden, regen_gpu

pytestmark = pytest.mark.gpu


def test_synth_matches_oracle(gpu, orc):
    """The HIP generator (bench data) is bit-identical to the oracle's generator."""
    for d, nc, n in [(768, 64, 3000), (100, 7, 500), (64, 1, 100)]:
        a = gpu.synth_vectors(d, nc, 1234, 10000, 0.5, 5, n)
        b = orc.synth_vectors(d, nc, 1234, 10000, 0.5, 5, n)
        assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"vectors d={d}"
        qa = gpu.synth_queries(d, nc, 1234, 10000, 0.5, n, 999, 0.1, 3, 50)
        qb = orc.synth_queries(d, nc, 1234, 10000, 0.5, n, 999, 0.1, 3, 50)
        assert np.array_equal(qa.view(np.uint16), qb.view(np.uint16)), f"queries d={d}"


@pytest.mark.parametrize("name", ["flat_ip_d768", "flat_l2_d64", "flat_ip_d100"])
def test_golden(gpu, name):
    g = load_golden(name)
    x, q = regen_gpu(gpu, g)
    ix = gpu.IndexFlat(g["d"], g["metric"])
    ix.add(x)
    assert ix.ntotal == g["n"] and ix.storage_dtype == "float16"
    D, I = ix.search(q, g["k"])
    assert_same_results(D, I, g["D"], g["I"], name)


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("nq", [3, 100, 200, 300])   # 3 -> streaming small-batch kernel, 100 -> 128x128 MFMA tiles, 200/300 -> 256x256 LDS-DMA tiles
def test_vs_oracle_both_kernels(gpu, orc, metric, nq):
    d, n, k = 768, 5000, 10
    x = orc.synth_vectors(d, 12, 21, 22, 0.5, 0, n)
    q = orc.synth_queries(d, 12, 21, 22, 0.5, n, 23, 0.1, 0, nq)
    ix = gpu.IndexFlat(d, metric)
    ix.add(x)
    D, I = ix.search(q, k)
    Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, metric)
    assert_same_results(D, I, Dr, Ir, f"flat metric={metric} nq={nq}")


def test_edge_cases(gpu):
    e = load_golden("edge_cases")
    ix = gpu.IndexFlatIP(32)
    # empty index: FAISS returns -1 / -inf
    D, I = ix.search(e["q"], 4)
    assert (I == -1).all() and np.isneginf(D).all()
    ix.add(e["x"])
    D, I = ix.search(e["q"], 8)
    assert_same_results(D, I, e["D"], e["I"], "duplicates")      # exact ties -> id ascending
    ix2 = gpu.IndexFlatIP(32)
    ix2.add(e["x"][:5])
    D, I = ix2.search(e["q"], 8)                                    # k > ntotal -> -1 / -inf padding
    assert_same_results(D, I, e["Dk"], e["Ik"], "k>ntotal")


def test_incremental_add_and_growth(gpu, orc):
    d, k = 64, 12
    x = orc.synth_vectors(d, 5, 31, 32, 0.5, 0, 3000)
    q = orc.synth_queries(d, 5, 31, 32, 0.5, 3000, 33, 0.1, 0, 50)
    a = gpu.IndexFlatIP(d)
    for lo, hi in [(0, 1), (1, 130), (130, 131), (131, 1500), (1500, 3000)]:   # forces several re-layouts
        a.add(x[lo:hi])
    b = gpu.IndexFlatIP(d)
    b.add(x)
    Da, Ia = a.search(q, k)
    Db, Ib = b.search(q, k)
    assert_same_results(Da, Ia, Db, Ib, "incremental")
    Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, 0)
    assert_same_results(Da, Ia, Dr, Ir, "incremental vs oracle")


def test_fp32_values_keep_fp32_storage(gpu, orc):
    rng = np.random.RandomState(3)
    d, n = 96, 2000
    x = rng.randn(n, d).astype(np.float32)          # not fp16-representable
    q = rng.randn(70, d).astype(np.float32)
    ix = gpu.IndexFlatIP(d)
    ix.add(x[:1000].astype(np.float16))             # starts as fp16 storage ...
    assert ix.storage_dtype == "float16"
    ix.add(x[1000:])                                # ... and is widened when a non-fp16 value arrives
    assert ix.storage_dtype == "float32"
    xs = np.concatenate([x[:1000].astype(np.float16).astype(np.float32), x[1000:]], 0)
    for qq in (q, q[:5]):
        D, I = ix.search(qq, 10)
        Dr, Ir = orc.flat_search(qq, xs, 10, 0)
        assert np.array_equal(I, Ir), "exact re-rank must recover the exact ids"
        assert np.allclose(D, Dr, rtol=0, atol=np.abs(Dr).max() * 2 ** -23)   # fp64 sum order: <= 1 ulp of fp32


@pytest.mark.parametrize("kind", ["flat", "ivfflat"])
@pytest.mark.parametrize("batch", [200, 7])
def test_lossy_fp16_scan_is_certified_or_rerun(gpu, orc, kind, batch):
    """fp32 rows and fp32 queries that fp16 cannot represent, on data where the fp16-MFMA scan CANNOT order the
    candidates (20000 near-duplicates that differ in bits fp16 drops): the rounding error exceeds the rank-k to
    rank-K' gap, so without the certificate the true top-k is silently lost.  With it every such query is
    flagged by k_finalize and re-run through the exact fp64 path: ids equal exact brute force."""
    rng = np.random.RandomState(11)
    d, n, k = 64, 20000, 10
    base = rng.randn(d).astype(np.float32)
    x = (base[None, :] + 1e-5 * rng.randn(n, d)).astype(np.float32)      # identical once rounded to fp16
    q = (base[None, :] + 0.01 * rng.randn(batch, d)).astype(np.float32)
    if kind == "flat":
        ix = gpu.IndexFlatIP(d)
    else:
        ix = gpu.IndexIVFFlat(None, d, 4, gpu.METRIC_INNER_PRODUCT)
        cen = rng.randn(4, d).astype(np.float32); cen[0] = base
        ix.set_centroids(cen / np.linalg.norm(cen, axis=1, keepdims=True))
        ix.nprobe = 4
    ix.add(x)
    assert ix.storage_dtype == "float32"
    Dr, Ir = orc.flat_search(q, x, k, 0)
    ix.set_param("profile", 1)
    D, I = ix.search(q, k)
    assert np.array_equal(I, Ir), "certified / re-run result must be the exact ids"
    assert np.allclose(D, Dr, rtol=0, atol=np.abs(Dr).max() * 2 ** -23)
    assert ix.get_timing("fallback_queries") > 0, "this data cannot be certified from an fp16 scan"
    ix.set_param("flat_cert", 0)                     # round-1 behaviour: the approximate top-K' is trusted
    D0, I0 = ix.search(q, k)
    assert not np.array_equal(I0, Ir), "the test data must actually defeat the uncertified scan"
    ix.set_param("flat_cert", 1)
    # well-separated data certifies without a single re-run (the fast path stays fast)
    x2 = orc.synth_vectors(d, 5, 41, 42, 0.5, 0, 5000)
    q2 = orc.synth_queries(d, 5, 41, 42, 0.5, 5000, 43, 0.1, 0, 50)
    jx = gpu.IndexFlatIP(d); jx.add(x2); jx.set_param("profile", 1)
    D2, I2 = jx.search(q2, k)
    assert_same_results(D2, I2, *orc.flat_search(q2.astype(np.float32), x2.astype(np.float32), k, 0), "certified fp16 data")
    assert jx.get_timing("fallback_queries") == 0


def test_explicit_ids_and_large_k(gpu, orc):
    d, n = 64, 4000
    x = orc.synth_vectors(d, 5, 41, 42, 0.5, 0, n)
    q = orc.synth_queries(d, 5, 41, 42, 0.5, n, 43, 0.1, 0, 40)
    ids = (np.arange(n, dtype=np.int64) * 7 + 1000003)[::-1].copy()
    ix = gpu.IndexFlatIP(d)
    ix.add_with_ids(x, ids)
    for k in (1, 100, 2048, 3000, 4096):            # 4096 = the reference backends' default k (flat.py:138)
        D, I = ix.search(q, k)
        Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, 0, ids=ids)
        assert_same_results(D, I, Dr, Ir, f"k={k}")
    assert ix._get("max_k") == 4096
    with pytest.raises(RuntimeError, match="4096"):
        ix.search(q, 4097)


def test_device_pointers_match_host_pointers(gpu, orc):
    import torch
    d, n = 768, 3000
    x = orc.synth_vectors(d, 5, 51, 52, 0.5, 0, n)
    q = orc.synth_queries(d, 5, 51, 52, 0.5, n, 53, 0.1, 0, 129)
    ix = gpu.IndexFlatIP(d)
    ix.add(torch.from_numpy(x).cuda())                      # add straight from HBM
    D, I = ix.search(q, 10)
    Dt, It = ix.search(torch.from_numpy(q).cuda(), 10)      # query + outputs resident in HBM
    assert Dt.is_cuda and It.is_cuda
    assert_same_results(Dt.cpu().numpy(), It.cpu().numpy(), D, I, "device vs host pointers")


def test_bad_arguments_raise(gpu):
    ix = gpu.IndexFlatIP(16)
    with pytest.raises(AssertionError):
        ix.add(np.zeros((4, 15), np.float32))
    with pytest.raises(RuntimeError):
        gpu.IndexIVFPQ(None, 16, 4, 3, 8, 0)        # d not a multiple of M
    with pytest.raises(RuntimeError, match="not.*implemented|only"):
        gpu.IndexIVFPQ(None, 16, 4, 4, 6, 0)        # nbits != 8
    iv = gpu.IndexIVFFlat(None, 16, 4, 0)
    with pytest.raises(RuntimeError, match="train"):
        iv.add(np.zeros((4, 16), np.float32))       # add before train


def test_flat_save_load_streams_in_chunks(gpu, tmp_path):
    """RSX1 save / load of a Flat index moves rows in 256k-row chunks (no whole-index temporaries: a 10M x 768 index is
    30 GB as fp32) — more than one chunk here, with and without explicit ids, storage dtype preserved."""
    d, n = 32, 300_000
    x = gpu.synth_vectors(d, 7, 3, 4, 0.5, 0, n)
    q = gpu.synth_queries(d, 7, 3, 4, 0.5, n, 5, 0.1, 0, 33)
    for with_ids in (False, True):
        ix = gpu.IndexFlatIP(d)
        ids = (np.arange(n, dtype=np.int64)[::-1] * 3 + 17).copy()
        if with_ids:
            ix.add_with_ids(x, ids)
        else:
            ix.add(x)
        D, I = ix.search(q, 10)
        path = str(tmp_path / f"flat_{int(with_ids)}.faiss")
        gpu.write_index(ix, path)
        jx = gpu.read_index(path)
        assert jx.ntotal == n and jx.storage_dtype == ix.storage_dtype == "float16"
        D2, I2 = jx.search(q, 10)
        assert np.array_equal(D, D2) and np.array_equal(I, I2)
        jx.add_with_ids(x[:5], ids[:5] + 10 ** 9) if with_ids else jx.add(x[:5])
        assert jx.ntotal == n + 5


@pytest.mark.parametrize("kind", ["flat", "ivfflat"])
def test_large_k_second_certificate_pass_settles_near_ties(gpu, orc, kind):
    """k_finalize at large k re-scores the k + max(8, k / 16) best approximate candidates first and certifies against that cut; a
    query it cannot clear gets the rest of its K' = 2048 state row re-scored and a second certificate (this path had no
    test).  Data that forces it: 1500 fp16 vectors within a hair of the query (their scores differ by less than the certificate's
    error bound, so the 1062nd approximate score cannot be separated from the 1000th exact one) and 18 500 far ones (the 2048th
    candidate is far below): the first certificate fails, the second clears — bit-equal to the oracle without an exact re-run."""
    rng = np.random.RandomState(5)
    d, n, near, k, nq = 256, 20000, 1500, 1000, 24
    c = rng.randn(d).astype(np.float32)
    x = (0.3 * rng.randn(n, d)).astype(np.float16)
    pos = rng.permutation(n)[:near]
    x[pos] = (c[None, :] + 0.002 * rng.randn(near, d)).astype(np.float16)
    q = (c[None, :] + 0.001 * rng.randn(nq, d)).astype(np.float16)
    if kind == "flat":
        ix = gpu.IndexFlatIP(d)
    else:
        ix = gpu.IndexIVFFlat(None, d, 4, gpu.METRIC_INNER_PRODUCT)
        cen = rng.randn(4, d).astype(np.float32); cen[0] = c
        ix.set_centroids(cen); ix.nprobe = 4
    ix.add(x)
    assert ix.storage_dtype == "float16"
    ix.set_param("profile", 1)
    D, I = ix.search(q, k)
    Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, 0)
    assert_same_results(D, I, Dr, Ir, f"{kind} k = 1000 over near-ties")
    assert ix.get_timing("fallback_queries") == 0, "the second certificate pass (K' = 2048) must clear what the first (1062 candidates) cannot"
    # the near cluster really is inside the first cut's error bound: the k-th and the 1062nd exact scores are closer than one part in 10^4
    Dn, _ = orc.flat_search(q.astype(np.float32), x.astype(np.float32), 1100, 0)
    assert np.all((Dn[:, k - 1] - Dn[:, 1061]) < 1e-4 * np.abs(Dn[:, k - 1]))


@pytest.mark.parametrize("cert", [1, 0])
def test_staged_filter_overflow_is_settled_exactly(gpu, orc, cert):
    """Flat, batch > 32: behind the threshold phase the rows are scanned by filtered GEMM launches whose epilogue appends the keys that
    beat the running K'-th key to a per-query candidate row (api_search.hip: search_batch, k_gemm.hip: k_flat_gemm2<true>).  Rows stored
    in ASCENDING score order defeat the threshold: every later row beats it and the candidate rows overflow.  With the certificate
    behind the search (default) the stages do not wait for their counts — k_finalize flags the queries and they join the exact re-run;
    with flat_cert = 0 the overflowed stage is redone chunk by chunk.  Either way: the oracle's ids and scores."""
    rng = np.random.RandomState(11)
    d, n, nq, k = 64, 400_000, 40, 10
    u = rng.randn(d).astype(np.float32); u /= np.linalg.norm(u)
    x = ((np.arange(n, dtype=np.float32)[:, None] / n) * u[None, :] + 0.001 * rng.randn(n, d).astype(np.float32)).astype(np.float16)
    q = (u[None, :] + 0.01 * rng.randn(nq, d).astype(np.float32)).astype(np.float16)
    ix = gpu.IndexFlatIP(d); ix.add(x)
    ix.set_param("flat_cert", cert); ix.set_param("profile", 1)
    D, I = ix.search(q, k)
    sample = [0, 7, 39]
    Dr, Ir = orc.flat_search(q[sample].astype(np.float32), x.astype(np.float32), k, 0)
    assert_same_results(D[sample], I[sample], Dr, Ir, f"flat_cert={cert}: overflowing candidate rows")
    if cert:
        assert ix.get_timing("fallback_queries") > 0, "the overflow must have been seen (k_finalize flags the query)"
    else:
        assert ix.get_timing("flat_filter_overflows") > 0, "the overflow must have been seen (per-stage count check)"
