"""GPU: the single-process multi-GPU handle (rsx_sharded_create) — ONE index object whose search spans
several shards, which is what the reference's one-call driver (src/search.py:296) and Indexer facade can reach.

On a 1-GPU box the shards share device 0 (the fan-out threads, per-shard streams, fan-in copies and the id-ordered merge are
all exercised; only the peer copies degenerate); with >= 2 visible GPUs the same checks run with one shard per device.
The bar is the single index's bits: a sharded handle is a partition of the SAME logical index."""
import os
import pickle

import numpy as np
import pytest

from util import assert_same_results

pytestmark = pytest.mark.gpu


def _data(gpu, d, n, nq, ncent=16):
    x = gpu.synth_vectors(d, ncent, 1234, 10000, 0.5, 0, n)
    q = gpu.synth_queries(d, ncent, 1234, 10000, 0.5, n, 999, 0.1, 0, nq)
    x[300:330] = x[5]          # exact score ties that straddle the shard boundaries of every add call
    return x, q


def _make(gpu, kind, d, nlist, M, devices=None):
    if kind == "flat":
        return gpu.IndexFlatIP(d, devices=devices)
    if kind == "ivfflat":
        return gpu.IndexIVFFlat(None, d, nlist, gpu.METRIC_INNER_PRODUCT, devices=devices)
    return gpu.IndexIVFPQ(None, d, nlist, M, 8, gpu.METRIC_INNER_PRODUCT, devices=devices)


def _devices(gpu, nshards):
    n = gpu.get_num_gpus()
    return [r % n for r in range(nshards)]


@pytest.mark.parametrize("kind", ["flat", "ivfflat", "ivfpq"])
@pytest.mark.parametrize("nshards", [2, 3])
def test_sharded_handle_equals_single_index(gpu, kind, nshards, tmp_path):
    d, n, nq, nlist, M, k = 96, 7000, 41, 8, 32, 20
    x, q = _data(gpu, d, n, nq)
    single = _make(gpu, kind, d, nlist, M)
    sh = _make(gpu, kind, d, nlist, M, devices=_devices(gpu, nshards))
    assert sh.nshards == nshards and single.nshards == 0
    if kind != "flat":
        single.train(x[:4000])
        sh.set_centroids(single.get_centroids())            # same parameters (sharded train is checked below)
        if kind == "ivfpq":
            sh.set_codebooks(single.get_codebooks())
        single.nprobe = sh.nprobe = 5
    assert sh.is_trained
    for c0 in range(0, n, 2600):                            # several add calls: each is cut into nshards pieces
        single.add(x[c0:c0 + 2600]); sh.add(x[c0:c0 + 2600])
    assert sh.ntotal == single.ntotal == n
    if kind != "flat":
        assert np.array_equal(sh.list_sizes(), single.list_sizes())
    Ds, Is = single.search(q, k)
    Dm, Im = sh.search(q, k)
    assert_same_results(Dm, Im, Ds, Is, f"sharded {kind} x{nshards}")
    # CUDA tensors in -> CUDA tensors out, same bits
    import torch
    Dt, It = sh.search(torch.from_numpy(q).cuda(), k)
    assert Dt.is_cuda and It.is_cuda
    assert_same_results(Dt.cpu().numpy(), It.cpu().numpy(), Ds, Is, f"sharded {kind} x{nshards}, device queries")
    D1, I1 = sh.search(q[3:4], 7)                           # one query, another k
    assert_same_results(D1, I1, *single.search(q[3:4], 7), f"sharded {kind} single query")
    # k larger than a shard's hits: -1 / -inf padding merges like the single index's
    if kind == "ivfpq":
        single.nprobe = sh.nprobe = 1
        assert_same_results(*sh.search(q, 50), *single.search(q, 50), "sharded ivfpq nprobe=1 k=50")
        single.nprobe = sh.nprobe = 5
    # persistence: manifest + one file per shard, back onto the same devices
    path = str(tmp_path / f"index_{kind}.faiss")
    gpu.write_index(sh, path)
    assert os.path.exists(path + ".shard0") and os.path.exists(path + f".shard{nshards - 1}")
    back = gpu.read_index(path, devices=_devices(gpu, nshards))
    assert back.nshards == nshards and back.ntotal == n
    if kind != "flat":
        back.nprobe = 5
    assert_same_results(*back.search(q, k), Ds, Is, f"sharded {kind} reloaded")
    back.add(x[:10])                                        # sequential ids continue after a reload
    assert back.ntotal == n + 10
    # knobs reach every shard; per-list import/export is refused with a clear error
    sh.set_param("query_batch", 7)
    assert_same_results(*sh.search(q, k), Ds, Is, f"sharded {kind} query_batch=7")
    if kind != "flat":
        with pytest.raises(RuntimeError):
            sh.get_list(0)
    sh.reset()
    assert sh.ntotal == 0


def test_sharded_train_copies_parameters(gpu):
    d, n, nlist, M = 64, 6000, 8, 16
    x, q = _data(gpu, d, n, 16)
    a = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0)
    b = gpu.IndexIVFPQ(None, d, nlist, M, 8, 0, devices=_devices(gpu, 2))
    a.train(x); b.train(x)
    assert np.array_equal(a.get_centroids(), b.get_centroids()) and np.array_equal(a.get_codebooks(), b.get_codebooks())
    a.add(x); b.add(x)
    a.nprobe = b.nprobe = 3
    assert_same_results(*b.search(q, 10), *a.search(q, 10), "sharded train")


class NS(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


@pytest.mark.parametrize("index_type", ["Flat", "IVFPQ"])
def test_indexer_facade_spans_devices(gpu, orc, tmp_path, index_type):
    """cfg.datastore.index.devices: the reference's Indexer(cfg).search — one call, all queries — over a sharded handle,
    built, saved (manifest + shard files under the reference's file names) and re-loaded through the facade."""
    import rsx
    from src.indicies.base import Indexer
    d, per, k = 768, 1800, 3
    embs = []

    def make_store(root):
        os.makedirs(os.path.join(root, "emb")); os.makedirs(os.path.join(root, "psg"))
        for s in range(2):
            e = orc.synth_vectors(d, 16, 11, 100 + s, 0.5, 0, per)
            if len(embs) < 2:
                embs.append(e)
            with open(os.path.join(root, "emb", f"passages_{s:02d}.pkl"), "wb") as f:
                pickle.dump((list(range(per)), e), f)
            with open(os.path.join(root, "psg", f"raw_passages-{s}-of-2.pkl"), "wb") as f:
                pickle.dump([{"text": f"s{s}c{c}"} for c in range(per)], f)

    def cfg(root, devices):
        idx = NS(index_type=index_type, index_shard_ids=[0, 1], projection_size=d, sample_train_size=3000, ncentroids=16,
                 probe=16, n_subquantizers=96, n_bits=8)
        if devices is not None:
            idx["devices"] = devices
        return NS(datastore=NS(embedding=NS(embedding_dir=os.path.join(root, "emb"), prefix="passages",
                                            passages_dir=os.path.join(root, "psg")), index=idx))
    r1, r2 = str(tmp_path / "one"), str(tmp_path / "many")
    make_store(r1); make_store(r2)
    q = orc.synth_queries(d, 16, 11, 100, 0.5, per, 7, 0.1, 0, 64)
    try:
        np.random.seed(5)                                   # the reference samples the training set unseeded
        one = Indexer(cfg(r1, None))
        assert one.datastore.index.nshards == 0
        s1 = one.search(q, k)
        np.random.seed(5)
        many = Indexer(cfg(r2, _devices(gpu, 2)))
        assert many.datastore.index.nshards == 2
        assert many.search(q, k) == s1
        again = Indexer(cfg(r2, _devices(gpu, 2)))          # load path: manifest + one file per shard
        assert again.datastore.index.nshards == 2 and again.search(q, k) == s1
    finally:
        rsx.set_default_devices(None)


def test_two_real_gpus(gpu):
    """One shard per visible device, device-resident queries on the last device (peer copies of queries and results)."""
    if gpu.get_num_gpus() < 2:
        pytest.skip("needs >= 2 visible GPUs (the 1-GPU box covers the same code with shards sharing device 0)")
    import torch
    ng = gpu.get_num_gpus()
    d, n, nq, k = 128, 40000, 64, 10
    x, q = _data(gpu, d, n, nq)
    single = gpu.IndexIVFPQ(None, d, 16, 32, 8, 0, device=0)
    sh = gpu.IndexIVFPQ(None, d, 16, 32, 8, 0, devices=list(range(ng)))
    single.train(x[:8000]); sh.set_centroids(single.get_centroids()); sh.set_codebooks(single.get_codebooks())
    single.add(x); sh.add(torch.from_numpy(x).to(f"cuda:{ng - 1}"))
    single.nprobe = sh.nprobe = 4
    Ds, Is = single.search(q, k)
    Dm, Im = sh.search(torch.from_numpy(q).to(f"cuda:{ng - 1}"), k)
    assert Dm.device.index == ng - 1
    assert_same_results(Dm.cpu().numpy(), Im.cpu().numpy(), Ds, Is, f"{ng} real GPUs")
