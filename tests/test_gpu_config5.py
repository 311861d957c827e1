"""GPU: BASELINE config 5 ("1B x 768 IVF-PQ sharded 8 x MI355X, RCCL top-k all-gather merge") scaled to ONE GPU.

Same index parameters and batch as the headline (M = 96, nbits 8, nlist 4096, nprobe 32, batch 1024, k = 10), EIGHT id-range
shards of 200k vectors each (1.6M vectors instead of 1B), through both multi-GPU forms:

  * `rsx_sharded_create` with 8 shards (one process; on a 1-GPU box the shards share cuda:0, on a node one per device),
  * `sharded.ShardedSearcher` with 8 real ranks (one process per shard; RCCL refuses several ranks per device, so on this
    box the ranks share cuda:0 and the ONE all_gather_into_tensor runs over gloo — the collective call and the merge are
    the ones `bench.py --gpus 8` issues over RCCL).

Bar: ids AND fp32 scores of all 1024 queries equal to the single index holding everything (a sharded index is a partition
of the same logical index) and to the CPU oracle run on the lists exported from the single index.
Reference semantics: per-shard search + score merge, src/search.py:282-303, 362-367; api/serve_main_node.py:150-163."""
import os
import socket

import numpy as np
import pytest

from util import assert_same_results

pytestmark = pytest.mark.gpu

D, NLIST, M, NPROBE, NQ, K = 768, 4096, 96, 32, 1024, 10
NSHARDS, PER_SHARD = 8, 200_000
N = NSHARDS * PER_SHARD
NCENT, SEED_C, SEED_X, SEED_Q = 4096, 1234, 10000, 999
N_TRAIN = 262_144           # 64 points per centroid: a test-sized training set (the bench uses 256 per centroid)


def _synth(rsx, torch, i0, n, dev):
    out = torch.empty((n, D), dtype=torch.float16, device=dev)
    rsx.synth_vectors(D, NCENT, SEED_C, SEED_X, 0.5, i0, n, out=out)
    return out


def _queries(rsx, torch, dev):
    q = torch.empty((NQ, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NCENT, SEED_C, SEED_X, 0.5, N, SEED_Q, 0.1, 0, NQ, out=q)
    return q


def _train_params(rsx, torch, dev):
    ix = rsx.IndexIVFPQ(None, D, NLIST, M, 8, rsx.METRIC_INNER_PRODUCT, device=0)
    ix.train(_synth(rsx, torch, 0, N_TRAIN, dev))
    return ix.get_centroids(), ix.get_codebooks()


def _new(rsx, cen, cb, devices=None):
    ix = rsx.IndexIVFPQ(None, D, NLIST, M, 8, rsx.METRIC_INNER_PRODUCT, device=None if devices else 0, devices=devices)
    ix.set_centroids(cen); ix.set_codebooks(cb)
    ix.nprobe = NPROBE
    return ix


def _oracle_results(orc, single, cen, cb, q32):
    """The CPU oracle on the probed lists exported from the single index (as bench.py's cpu_baseline leg does)."""
    ls = single.list_sizes()
    pid, _ = orc.coarse_probe(cen, q32, NPROBE)
    need = np.unique(pid[pid >= 0])
    lens = np.zeros(NLIST, dtype=np.int64); lens[need] = ls[need]
    off = np.zeros(NLIST + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])

    class LM:
        pass
    lm = LM()
    lm.list_off = off
    lm.payload = np.empty((int(off[-1]), M), np.uint8)
    lm.ids = np.empty(int(off[-1]), np.int64)
    for l in need:
        c, i = single.get_list(int(l))
        lm.payload[off[l]:off[l + 1]] = c; lm.ids[off[l]:off[l + 1]] = i
    return orc.ivfpq_search(cen, cb, lm, q32, NPROBE, K)


def test_config5_eight_shards_one_handle(gpu, orc):
    import torch
    dev = torch.device("cuda", 0)
    cen, cb = _train_params(gpu, torch, dev)
    ndev = gpu.get_num_gpus()
    single = _new(gpu, cen, cb)
    sh = _new(gpu, cen, cb, devices=[r % ndev for r in range(NSHARDS)])
    assert sh.nshards == NSHARDS
    for r in range(NSHARDS):                      # shard r of the handle receives piece r of EVERY add call
        x = _synth(gpu, torch, r * PER_SHARD, PER_SHARD, dev)
        single.add(x); sh.add(x)
    assert single.ntotal == sh.ntotal == N
    assert np.array_equal(single.list_sizes(), sh.list_sizes())
    q = _queries(gpu, torch, dev)
    Ds, Is = single.search(q, K)
    Dm, Im = sh.search(q, K)
    Ds, Is, Dm, Im = Ds.cpu().numpy(), Is.cpu().numpy(), Dm.cpu().numpy(), Im.cpu().numpy()
    assert_same_results(Dm, Im, Ds, Is, "config 5: 8-shard handle vs single index")
    Do, Io = _oracle_results(orc, single, cen, cb, q.cpu().numpy().astype(np.float32))
    assert_same_results(Ds, Is, Do, Io, "config 5: single index vs CPU oracle, all 1024 queries")
    assert (Is >= 0).all() and len(np.unique(Is // PER_SHARD)) == NSHARDS      # results really come from every shard
    # host queries in / host results out take the same path
    Dh, Ih = sh.search(q.cpu().numpy(), K)
    assert_same_results(Dh, Ih, Ds, Is, "config 5: 8-shard handle, host queries")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, params_path, ret):
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
    ndev = rsx.get_num_gpus()
    dev = torch.device("cuda", rank % ndev)
    torch.cuda.set_device(dev)
    p = np.load(params_path)
    local = rsx.IndexIVFPQ(None, D, NLIST, M, 8, rsx.METRIC_INNER_PRODUCT, device=rank % ndev)
    local.set_centroids(p["cen"]); local.set_codebooks(p["cb"])
    local.nprobe = NPROBE
    lo, hi = shard_range(N, rank, world)
    assert hi - lo == PER_SHARD
    x = torch.empty((hi - lo, D), dtype=torch.float16, device=dev)
    rsx.synth_vectors(D, NCENT, SEED_C, SEED_X, 0.5, lo, hi - lo, out=x, device=rank % ndev)
    local.add(x)
    q = torch.empty((NQ, D), dtype=torch.float16, device=dev)
    rsx.synth_queries(D, NCENT, SEED_C, SEED_X, 0.5, N, SEED_Q, 0.1, 0, NQ, out=q, device=rank % ndev)
    Dm, Im = ShardedSearcher(local, id_offset=lo).search(q.cpu().numpy(), K)     # same batch on every rank, merged everywhere
    ret[rank] = (np.asarray(Dm), np.asarray(Im))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_eight_ranks_sharded_searcher(gpu, tmp_path):
    import torch
    import torch.multiprocessing as mp
    dev = torch.device("cuda", 0)
    cen, cb = _train_params(gpu, torch, dev)
    single = _new(gpu, cen, cb)
    for r in range(NSHARDS):
        single.add(_synth(gpu, torch, r * PER_SHARD, PER_SHARD, dev))
    Ds, Is = single.search(_queries(gpu, torch, dev), K)
    Ds, Is = Ds.cpu().numpy(), Is.cpu().numpy()
    del single
    torch.cuda.empty_cache()
    params = str(tmp_path / "params.npz")
    np.savez(params, cen=cen, cb=cb)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_rank, args=(NSHARDS, _free_port(), params, ret), nprocs=NSHARDS, join=True)
        got = dict(ret)
    assert set(got) == set(range(NSHARDS))
    for r in range(NSHARDS):                      # every rank holds the merged result, and it is the single index's
        assert_same_results(got[r][0], got[r][1], Ds, Is, f"config 5: rank {r} of 8, merged result vs single index")


def test_headline_size_100M_parity_and_invariants(gpu, orc):
    """BASELINE config 4 at its FULL size — 100M x 768, IVF-PQ M = 96, nlist 4096, nprobe 32, batch 1024, k = 10 — as a test
    (full-size parity used to rest on the bench line alone).  The lists are ~24k vectors long here (366 at the
    1.5M test sizes): whole-list tiles, multi-tile lists, sibling-group joins and the work stealing all run at their real
    shapes.  Checks: a sample of queries bit-equal to the CPU oracle on the exported probed lists, batch-split invariance of
    the whole batch, k = 100 (the 16k-vector pre-pass sample and the emission path) against the oracle, and no exact re-runs."""
    import torch
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * (1 << 30):
        pytest.skip("needs ~25 GB of free HBM")
    n, nq, k = 100_000_000, 1024, 10
    ix = gpu.IndexIVFPQ(None, D, NLIST, M, 8, gpu.METRIC_INNER_PRODUCT, device=0)
    nt = 256 * NLIST
    xt = torch.empty((nt, D), dtype=torch.float16, device=dev)
    stride = n // nt
    for b in range(0, nt, 4096):
        gpu.synth_vectors(D, NCENT, SEED_C, SEED_X, 0.5, (b * stride) % (n - 4096), 4096, out=xt[b:b + 4096])
    ix.train(xt); del xt
    ix.nprobe = NPROBE
    buf = torch.empty((1_000_000, D), dtype=torch.float16, device=dev)
    for c0 in range(0, n, 1_000_000):
        gpu.synth_vectors(D, NCENT, SEED_C, SEED_X, 0.5, c0, 1_000_000, out=buf); ix.add(buf)
    del buf
    assert ix.ntotal == n
    q = torch.empty((nq, D), dtype=torch.float16, device=dev)
    gpu.synth_queries(D, NCENT, SEED_C, SEED_X, 0.5, n, SEED_Q, 0.1, 0, nq, out=q)
    ix.set_param("profile", 1)
    Dg, Ig = ix.search(q, k)
    assert ix.get_timing("fallback_queries") <= 2                       # the fast scan certifies (nearly) every query at this size
    for s in range(0, nq, 256):                                         # batch decomposition must be invisible
        Ds, Is = ix.search(q[s:s + 256], k)
        assert torch.equal(Ds, Dg[s:s + 256]) and torch.equal(Is, Ig[s:s + 256])
    sample = [0, 1, 255, 256, 511, 640, 777, 1023]
    qs = q[sample].cpu().numpy().astype(np.float32)
    cen, cb = ix.get_centroids(), ix.get_codebooks()
    ls = ix.list_sizes()
    pid, _ = orc.coarse_probe(cen, qs, NPROBE)
    need = np.unique(pid[pid >= 0])
    lens = np.zeros(NLIST, dtype=np.int64); lens[need] = ls[need]
    off = np.zeros(NLIST + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])

    class LM:
        pass
    lm = LM()
    lm.list_off = off
    lm.payload = np.empty((int(off[-1]), M), np.uint8)
    lm.ids = np.empty(int(off[-1]), np.int64)
    for l in need:
        c, i = ix.get_list(int(l))
        lm.payload[off[l]:off[l + 1]] = c; lm.ids[off[l]:off[l + 1]] = i
    Do, Io = orc.ivfpq_search(cen, cb, lm, qs, NPROBE, k)
    assert_same_results(Dg.cpu().numpy()[sample], Ig.cpu().numpy()[sample], Do, Io, "100M IVF-PQ vs CPU oracle (sample of 8 queries)")
    D1, I1 = ix.search(q[sample], 100)
    Do1, Io1 = orc.ivfpq_search(cen, cb, lm, qs, NPROBE, 100)
    assert_same_results(D1.cpu().numpy(), I1.cpu().numpy(), Do1, Io1, "100M IVF-PQ k = 100 vs CPU oracle")
