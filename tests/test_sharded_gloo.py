"""CPU: the N>1 search path with world_size 2 over gloo (one process per shard).

Each rank owns a contiguous shard, searches it with the test-double engine, and ShardedSearcher
exchanges candidates with ONE all-gather and merges.  The merged result must equal a search over
the union."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


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
    import torch.distributed as dist
    import fake_engine
    from oracle import oracle as o
    from sharded import ShardedSearcher, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, k = 1501, 48, 6
    x = o.synth_vectors(d, 9, 3, 4, 0.5, 0, n)
    x[700] = x[20]                     # a cross-shard exact tie: ids 20 (shard 0) and 700 (shard 1)
    q = np.concatenate([x[20:21], o.synth_queries(d, 9, 3, 4, 0.5, n, 5, 0.1, 0, 7)], 0)
    lo, hi = shard_range(n, rank, world)
    local = fake_engine.IndexFlatIP(d)
    local.add(x[lo:hi])
    D, I = ShardedSearcher(local, id_offset=lo).search(q, k)
    Dref, Iref = o.flat_search(q.astype(np.float32), x.astype(np.float32), k, 0)
    ok = np.array_equal(I, Iref) and np.array_equal(D, Dref) and I[0, 0] == 20 and I[0, 1] == 700
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_matches_union():
    world = 2
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
