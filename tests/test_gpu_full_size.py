"""GPU: BASELINE configs 2 and 3 at their FULL sizes, as tests (config 4 has
tests/test_gpu_config5.py::test_headline_size_100M_parity_and_invariants).

  * config 2 — 10M x 768 Flat, batch 1024, both metrics (`FlatIndexer.search`, reference src/indicies/flat.py:138-141):
    k_flat_gemm2 over 10M rows, the staged filtered GEMM launches, the certificate;
  * config 3 — 100M x 768 IVF-Flat, nlist 4096, nprobe 32 (`IVFFlatIndexer.search`, src/indicies/ivf_flat.py:224-227):
    153.6 GB of fp16 rows in one allocation, ~24k-row lists, the LDS-DMA list scan with its in-kernel candidate filter.

Each: eight queries of the timed batch bit-equal (ids AND fp32 scores) to the CPU oracle — for Flat the oracle streams the
regenerated database chunk by chunk and merges, for IVF-Flat it scans the probed lists exported from the index —, the result
of the whole batch invariant under a split into four batches, and ZERO certificate fallbacks (the fast path must settle every
query of this data by itself).  Synthetic data as SURVEY 8(d) prescribes (the bench's generator and seeds)."""
import numpy as np
import pytest

from util import assert_same_results

pytestmark = pytest.mark.gpu

D, NC, SC, SX, SQ = 768, 4096, 1234, 10000, 999
SAMPLE = [0, 1, 255, 256, 511, 640, 777, 1023]


def _queries(gpu, torch, dev, n, nq):
    q = torch.empty((nq, D), dtype=torch.float16, device=dev)
    gpu.synth_queries(D, NC, SC, SX, 0.5, n, SQ, 0.1, 0, nq, out=q)
    return q


def _split_invariant(torch, ix, q, k, Dg, Ig):
    for s in range(0, q.shape[0], 256):
        Ds, Is = ix.search(q[s:s + 256], k)
        assert torch.equal(Ds, Dg[s:s + 256]) and torch.equal(Is, Ig[s:s + 256]), f"batch split at {s} changes the result"


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_config2_flat_10M_batch_1024(gpu, orc, metric):
    import torch
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * (1 << 30):
        pytest.skip("needs ~20 GB of free HBM")
    n, nq, k = 10_000_000, 1024, 10
    mcode = 1 if metric == "l2" else 0
    ix = gpu.IndexFlat(D, gpu.METRIC_L2 if mcode else gpu.METRIC_INNER_PRODUCT)
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, n, buf.shape[0]):
        gpu.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf); ix.add(buf)
    assert ix.ntotal == n and ix.storage_dtype == "float16"
    q = _queries(gpu, torch, dev, n, nq)
    ix.set_param("profile", 1)
    Dg, Ig = ix.search(q, k)
    assert ix.get_timing("fallback_queries") == 0, "the certificate must clear every query of the batch"
    _split_invariant(torch, ix, q, k, Dg, Ig)
    # the oracle streams the database: exact top-k of every 1M-row chunk, merged
    qs = q[SAMPLE].cpu().numpy().astype(np.float32)
    best = None
    for c0 in range(0, n, buf.shape[0]):
        gpu.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf)
        Dc, Ic = orc.flat_search(qs, buf.cpu().numpy().astype(np.float32), k, mcode)
        Ic = Ic + c0
        best = (Dc, Ic) if best is None else orc.merge_topk(np.stack([best[0], Dc]), np.stack([best[1], Ic]), mcode)
    assert_same_results(Dg.cpu().numpy()[SAMPLE], Ig.cpu().numpy()[SAMPLE], best[0], best[1], f"10M Flat {metric} vs CPU oracle (8 queries)")
    # the reference's n_docs on the same index (ric/conf/default.yaml: 100 ... 1000), oracle on the same eight queries: k = 100 as a batch of
    # eight (the list-scan form), k = 1000 as the WHOLE batch (round 6: the staged filtered GEMM launches with their queued epilogue, six stages
    # that do not wait for their counts, the two-pass certificate of k_finalize) — the oracle's top 100 are the first 100 of its top 1000
    D1, I1 = ix.search(q[SAMPLE], 100)
    Dk, Ik = ix.search(q, 1000)
    assert ix.get_timing("fallback_queries") == 0 and ix.get_timing("flat_filter_overflows") == 0, "k = 1000: no re-run, no overflowed stage"
    best = None
    for c0 in range(0, n, buf.shape[0]):
        gpu.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf)
        Dc, Ic = orc.flat_search(qs, buf.cpu().numpy().astype(np.float32), 1000, mcode)
        Ic = Ic + c0
        best = (Dc, Ic) if best is None else orc.merge_topk(np.stack([best[0], Dc]), np.stack([best[1], Ic]), mcode)
    assert_same_results(D1.cpu().numpy(), I1.cpu().numpy(), best[0][:, :100], best[1][:, :100], f"10M Flat {metric} k = 100 vs CPU oracle")
    assert_same_results(Dk.cpu().numpy()[SAMPLE], Ik.cpu().numpy()[SAMPLE], best[0], best[1], f"10M Flat {metric} k = 1000 (whole batch) vs CPU oracle")


def test_config3_ivfflat_100M_nlist4096_nprobe32(gpu, orc):
    import torch
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~170 GB of free HBM (153.6 GB of fp16 rows + ids + workspaces)")
    n, nlist, nprobe, nq, k = 100_000_000, 4096, 32, 1024, 10
    ix = gpu.IndexIVFFlat(None, D, nlist, gpu.METRIC_INNER_PRODUCT)
    nt = 256 * nlist
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    gpu.synth_vectors(D, NC, SC, SX, 0.5, 0, nt, out=xt)
    ix.train(xt); del xt
    ix.nprobe = nprobe
    # pass 1 counts the list sizes (quantizer.assign) and reserves them exactly — a re-layout of a 153.6 GB index cannot hold two
    # copies in 288 GB —, pass 2 adds (tools/bench_configs.py builds the bench's config-3 leg the same way)
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    counts = np.zeros(nlist, dtype=np.int64)
    for c0 in range(0, n, buf.shape[0]):
        gpu.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf)
        counts += np.bincount(ix.assign(buf), minlength=nlist)
    ix.reserve_lists(counts)
    for c0 in range(0, n, buf.shape[0]):
        gpu.synth_vectors(D, NC, SC, SX, 0.5, c0, buf.shape[0], out=buf); ix.add(buf)
    del buf
    assert ix.ntotal == n and ix.storage_dtype == "float16"
    assert np.array_equal(ix.list_sizes(), counts)
    q = _queries(gpu, torch, dev, n, nq)
    ix.set_param("profile", 1)
    Dg, Ig = ix.search(q, k)
    assert ix.get_timing("fallback_queries") == 0, "the certificate must clear every query of the batch"
    _split_invariant(torch, ix, q, k, Dg, Ig)
    qs = q[SAMPLE].cpu().numpy().astype(np.float32)
    cen = ix.get_centroids()
    pid, _ = orc.coarse_probe(cen, qs, nprobe)
    need = np.unique(pid[pid >= 0])
    lens = np.zeros(nlist, np.int64); lens[need] = counts[need]
    off = np.zeros(nlist + 1, np.int64); np.cumsum(lens, out=off[1:])

    class LM:
        pass
    lm = LM(); lm.list_off = off
    lm.payload = np.empty((int(off[-1]), D), np.float32)
    lm.ids = np.empty(int(off[-1]), np.int64)
    for l in need:
        v, i = ix.get_list(int(l))
        lm.payload[off[l]:off[l + 1]] = v; lm.ids[off[l]:off[l + 1]] = i
    Do, Io = orc.ivfflat_search(0, cen, lm, qs, nprobe, k)
    assert_same_results(Dg.cpu().numpy()[SAMPLE], Ig.cpu().numpy()[SAMPLE], Do, Io, "100M IVF-Flat vs CPU oracle (8 queries)")
    D1, I1 = ix.search(q[SAMPLE], 100)
    Do1, Io1 = orc.ivfflat_search(0, cen, lm, qs, nprobe, 100)
    assert_same_results(D1.cpu().numpy(), I1.cpu().numpy(), Do1, Io1, "100M IVF-Flat k = 100 vs CPU oracle")
    # ... and the reference's n_docs = 1000 as the WHOLE batch: threshold sample of each query's two closest lists, the filtered list scan,
    # the two-pass certificate — no candidate row may overflow and no query may need the exact re-run on this data
    Dk, Ik = ix.search(q, 1000)
    assert ix.get_timing("fallback_queries") == 0 and ix.get_timing("ivf_filter_overflow_queries") == 0
    Dok, Iok = orc.ivfflat_search(0, cen, lm, qs, nprobe, 1000)
    assert_same_results(Dk.cpu().numpy()[SAMPLE], Ik.cpu().numpy()[SAMPLE], Dok, Iok, "100M IVF-Flat k = 1000 (whole batch) vs CPU oracle")
