"""GPU: the one-process-per-GPU search path (sharded.ShardedSearcher) executed by TWO real ranks through librsx.

The GPU box has one device and RCCL refuses two ranks on one GPU, so the two processes share cuda:0 and exchange their
candidates over gloo (the collective is the same ONE all_gather_into_tensor of a packed [2, nq, k] block; the RCCL form of
it runs with world_size 1 in test_gpu_ivf.py and with N ranks in `bench.py --gpus N`).  What this covers that the CPU gloo
test cannot: every rank building its id-range shard of ONE logical IVF-PQ / Flat index with the HIP engine (shared trained
parameters), searching it, and the merged result being the single index's, bit for bit — including a cross-shard exact
tie and k larger than one shard's hits."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "retrieval-scaling_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import rsx
    from sharded import ShardedSearcher, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d, n, nq, k, nlist, M = 96, 9000, 33, 12, 8, 32
    x = rsx.synth_vectors(d, 16, 77, 5000, 0.5, 0, n)
    x[6100:6104] = x[40]                    # exact score ties across the two shards (ids 40 | 6100..6103)
    q = rsx.synth_queries(d, 16, 77, 5000, 0.5, n, 31, 0.1, 0, nq)
    q[0] = x[40]
    lo, hi = shard_range(n, rank, world)
    ok = {}
    for kind in ("flat", "ivfpq"):
        if kind == "flat":
            full, local = rsx.IndexFlatIP(d), rsx.IndexFlatIP(d)
        else:
            full = rsx.IndexIVFPQ(None, d, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
            local = rsx.IndexIVFPQ(None, d, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
            full.train(x[:4000])
            local.set_centroids(full.get_centroids()); local.set_codebooks(full.get_codebooks())   # one logical index
            full.nprobe = local.nprobe = 4
        full.add(x)
        local.add(x[lo:hi])
        searcher = ShardedSearcher(local, id_offset=lo)
        for kk in (k, 60):
            D, I = searcher.search(q, kk)                       # numpy in -> numpy out, merged on every rank
            Dr, Ir = full.search(q, kk)
            ok[f"{kind}_k{kk}"] = bool(np.array_equal(I, Ir) and np.array_equal(D, Dr))
        ok[f"{kind}_tie"] = bool(list(searcher.search(q[:1], 5)[1][0]) == list(full.search(q[:1], 5)[1][0]))
        if kind == "ivfpq":
            # LIST shards + the threshold exchange: rank r keeps the lists l % 2 == r of the same add stream; the search runs in
            # two calls (rsx_search_prepass / rsx_search_scan) with ONE all-reduce(MAX) of the threshold keys in between
            ls = rsx.IndexIVFPQ(None, d, nlist, M, 8, rsx.METRIC_INNER_PRODUCT)
            ls.set_centroids(full.get_centroids()); ls.set_codebooks(full.get_codebooks())
            ls.set_param("add_list_mod", world); ls.set_param("add_list_rem", rank)
            ls.nprobe = 4
            ls.add(x)
            ls.set_param("profile", 1)
            qd = torch.from_numpy(q).cuda()
            Dl, Il = ShardedSearcher(ls, id_offset=0, exchange_thresholds=True).search(qd, k)
            Dr, Ir = full.search(q, k)
            ok["ivfpq_list_shards_exchanged"] = bool(np.array_equal(Il.cpu().numpy(), Ir) and np.array_equal(Dl.cpu().numpy(), Dr))
            ok["ivfpq_list_shards_no_exact_rerun"] = bool(ls.get_timing("fallback_queries") == 0)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_two_real_ranks_share_one_gpu(gpu):
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        got = dict(ret)
    assert set(got) == {0, 1}
    for r in (0, 1):
        assert all(got[r].values()), (r, got[r])
