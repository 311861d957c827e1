"""Randomised parity sweep: many small (shape, nprobe, k, batch) draws per index type against the oracle.
Sizes are chosen to cross the engine's internal switches: pre-pass prefix shorter/longer than the closest
list, K' above/below the filter guards, one-query batches, empty lists, k > candidates (padding)."""
import numpy as np
import pytest

from util import assert_same_results

pytestmark = pytest.mark.gpu


def _data(orc, rng, d, n, nq, ncent):
    x = orc.synth_vectors(d, ncent, int(rng.randint(1, 1 << 20)), int(rng.randint(1, 1 << 20)), 0.5, 0, n)
    q = orc.synth_queries(d, ncent, 11, 12, 0.5, n, int(rng.randint(1, 1 << 20)), 0.1, 0, nq)
    return x, q


@pytest.mark.parametrize("seed", range(10))
def test_random_ivfpq(gpu, orc, seed):
    """Round 6: the draw also covers the metric (inner product / L2), the code layout (granule / rotated / sliced where they apply) and the
    knobs that pick between scan and finalize kernels (eight / four queries per gather, the row-major code copy, gather + select)."""
    rng = np.random.RandomState(100 + seed)
    d, M = [(768, 96), (96, 96), (128, 16), (64, 32), (768, 96), (768, 96), (256, 64), (768, 96), (192, 96), (512, 128)][seed]
    nlist = int(rng.choice([3, 8, 32]))
    n = int(rng.choice([700, 5000, 16000]))
    nq = int(rng.choice([1, 5, 70])) if seed < 4 else int(rng.choice([1, 9, 70, 140]))
    metric = 0 if seed < 4 else int(rng.randint(0, 2))
    x, q = _data(orc, rng, d, n, nq, nlist)
    x32 = x.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 2, 7)
    a, _ = orc.assign_ip(cen, x32)
    res = orc.residuals(cen, x32, a)
    cb = orc.pq_train(res[:2000], M, 1, 7)
    lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, res), nlist)
    ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, metric)
    knobs = {}
    if seed >= 4:
        layouts = [0] + ([1] if M in (16, 32, 64, 96, 128) else []) + ([2] if M == 96 else [])
        knobs = {"pq_layout": int(rng.choice(layouts)), "pq_q8": int(rng.choice([0, 1, 2])), "pq_plain_codes": int(rng.randint(0, 2)),
                 "pq_gather": int(rng.randint(0, 2)), "pq_prepass4": int(rng.choice([0, 1, 2]))}
        for name, v in knobs.items():
            ix.set_param(name, v)
    ix.set_centroids(cen); ix.set_codebooks(cb)
    for c0 in range(0, n, 7001):                      # several add calls: list growth / re-layout
        ix.add(x[c0:c0 + 7001])
    for nprobe, k in [(1, 1), (2, 10), (nlist, 10), (max(2, nlist // 2), 300), (nlist, 4096)]:
        ix.nprobe = nprobe
        D, I = ix.search(q, k)
        Dr, Ir = orc.ivfpq_search(cen, cb, lm, q.astype(np.float32), nprobe, k, metric=metric)
        assert_same_results(D, I, Dr, Ir, f"ivfpq seed={seed} d={d} M={M} metric={metric} nlist={nlist} n={n} nq={nq} nprobe={nprobe} k={k} {knobs}")


@pytest.mark.parametrize("seed", range(4))
def test_random_ivfflat(gpu, orc, seed):
    rng = np.random.RandomState(200 + seed)
    d = int(rng.choice([64, 128, 768, 100]))          # 100: row stride padded to 128
    nlist = int(rng.choice([2, 8, 16]))
    n = int(rng.choice([600, 4000, 20000]))
    nq = int(rng.choice([1, 17, 90]))
    metric = seed % 2
    x, q = _data(orc, rng, d, n, nq, nlist)
    x32 = x.astype(np.float32)
    cen = orc.kmeans(0, x32, nlist, 3, 7)
    a, _ = orc.assign_ip(cen, x32)
    lm = orc.ListMajor(a, np.arange(n), x32, nlist)
    ix = gpu.IndexIVFFlat(None, d, nlist, metric)
    ix.set_centroids(cen)
    ix.add(x)
    for filt in (1, 2):
        ix.set_param("ivf_filter", filt)
        for nprobe, k in [(1, 1), (2, 10), (nlist, 100), (nlist, 1000)]:
            ix.nprobe = nprobe
            D, I = ix.search(q, k)
            Dr, Ir = orc.ivfflat_search(metric, cen, lm, q.astype(np.float32), nprobe, k)
            what = f"ivfflat seed={seed} d={d} nlist={nlist} n={n} nq={nq} metric={metric} nprobe={nprobe} k={k} filter={filt}"
            assert np.array_equal(I, Ir), what
            fin = np.isfinite(Dr)
            assert np.array_equal(np.isfinite(D), fin) and np.allclose(D[fin], Dr[fin], rtol=0, atol=max(1e-30, np.abs(Dr[fin]).max() * 2 ** -22)), what


@pytest.mark.parametrize("seed", range(3))
def test_random_flat(gpu, orc, seed):
    rng = np.random.RandomState(300 + seed)
    d = int(rng.choice([64, 768, 200]))
    n = int(rng.choice([50, 3000, 70000]))            # 70000 > 65536: the filtered single-launch GEMM path
    nq = int(rng.choice([1, 40, 129, 300]))
    metric = seed % 2
    x, q = _data(orc, rng, d, n, nq, 9)
    ix = gpu.IndexFlat(d, metric)
    ix.add(x)
    for k in (1, 10, 200):
        D, I = ix.search(q, k)
        Dr, Ir = orc.flat_search(q.astype(np.float32), x.astype(np.float32), k, metric)
        assert_same_results(D, I, Dr, Ir, f"flat seed={seed} d={d} n={n} nq={nq} metric={metric} k={k}")


@pytest.mark.parametrize("seed", range(8))
def test_random_coarse_quantiser(gpu, orc, seed):
    """Round 6: the fast coarse quantiser (fp16 MFMA scores + exact chains of the candidates) and the matrix-core table build over random
    shapes: list counts on both sides of the register-resident key row sizes (1024 / 4096), dimensions that are not multiples of 4 / 64 / 128,
    nprobe from 1 to the 48 the fast form serves, fp16 and fp32 queries, batch sizes around the 32-query tiles, IVF-Flat and IVF-PQ.  The
    centroids are data points (a query near one of them ties nothing) and partly scaled copies of each other (near-ties in the approximate
    scores); every result must be the oracle's, with at most a few queries taking the exact re-run."""
    rng = np.random.RandomState(300 + seed)
    d = int([64, 100, 768, 96, 256, 130, 32, 512][seed])
    nlist = int([300, 1025, 4096, 50, 2000, 1024, 4500, 97][seed])
    nprobe = int(min(nlist - 1, [32, 48, 32, 7, 1, 20, 40, 48][seed]))
    n = int(max(4 * nlist, 20000))
    nq = int([33, 128, 200, 64, 97, 32, 130, 70][seed])
    x, q = _data(orc, rng, d, n, nq, 64)
    x32 = x.astype(np.float32)
    cen = x32[rng.choice(n, nlist, replace=False)].copy()
    dup = rng.choice(nlist, nlist // 10, replace=False)
    cen[dup] = cen[(dup + 1) % nlist] * (1.0 + 1e-4 * rng.randn(len(dup), 1).astype(np.float32))
    a, _ = orc.assign_ip(cen, x32)
    pq = seed % 2 == 0 and d % 8 == 0
    if pq:
        M = d // 8
        if M not in (4, 8, 12, 32, 64, 96): M = 8 if d % 8 == 0 else 4
        res = orc.residuals(cen, x32, a)
        cb = orc.pq_train(res[:3000], M, 1, 7)
        lm = orc.ListMajor(a, np.arange(n), orc.pq_encode(cb, res), nlist)
        ix = gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT)
        ix.set_centroids(cen); ix.set_codebooks(cb)
    else:
        lm = orc.ListMajor(a, np.arange(n), x32, nlist)
        ix = gpu.IndexIVFFlat(None, d, nlist, gpu.METRIC_INNER_PRODUCT)
        ix.set_centroids(cen)
    ix.add(x); ix.nprobe = nprobe
    ix.set_param("profile", 1)
    for qq, what in ((q, "fp16"), (q.astype(np.float32) * (1.0 + 1e-3 * rng.randn(nq, 1).astype(np.float32)), "fp32")):
        qf = qq.astype(np.float32)
        for k in (1, 10):
            D, I = ix.search(qq, k)
            if pq:
                Dr, Ir = orc.ivfpq_search(cen, cb, lm, qf, nprobe, k)
            else:
                Dr, Ir = orc.ivfflat_search(0, cen, lm, qf, nprobe, k)
            assert_same_results(D, I, Dr, Ir, f"seed={seed} {'ivfpq' if pq else 'ivfflat'} d={d} nlist={nlist} nprobe={nprobe} nq={nq} k={k} {what} queries")
    assert ix.get_timing("coarse_redo_queries") <= 0.2 * 4 * nq, "the fast coarse quantiser must settle nearly every query by itself"
