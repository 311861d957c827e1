"""Multi-shard merge beyond one launch (round 5): nshards * k > 16384 keys per query merge in ROUNDS over groups of consecutive
shards (k_select.hip: merge_any).  The rule is the reference's (src/search.py:362-367: concat in shard order, stable sort by
score, keep k) — sharded.merge_topk_host restates it and is pinned to the reference's own rerank_elements in
tests/test_host_logic.py — and the rounds must give that result bit for bit, for plain [nshards, nq, k] arrays, for the packed
all-gather buffer of the RCCL path, and with whole runs of equal scores crossing shard and group boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_case(rng, ns, nq, k, ties):
    if ties:      # few distinct scores: every query has long runs of ties across shards
        D = rng.randint(0, 7, size=(ns, nq, k)).astype(np.float32)
    else:
        D = rng.randn(ns, nq, k).astype(np.float32)
    D = -np.sort(-D, axis=2)                                   # every shard's list is sorted, best first
    I = rng.randint(0, 1 << 40, size=(ns, nq, k)).astype(np.int64)
    # padding at the tail of some lists (a shard that held fewer than k vectors)
    for s in range(ns):
        for q in range(nq):
            if rng.rand() < 0.3:
                cut = rng.randint(0, k + 1)
                I[s, q, cut:] = -1; D[s, q, cut:] = -np.inf
    return D, I


@pytest.mark.parametrize("ns,k", [(8, 4096), (5, 4096), (3, 8192), (33, 1000), (8, 2048), (8, 2049), (64, 300), (1, 16384), (1, 10000)])
@pytest.mark.parametrize("ties", [False, True])
def test_merge_in_rounds_matches_reference_rule(gpu, ns, k, ties):
    import torch
    import sharded
    rng = np.random.RandomState(ns * 10007 + k)
    nq = 5
    D, I = make_case(rng, ns, nq, k, ties)
    Dr, Ir = sharded.merge_topk_host(D, I, 0)
    Dg, Ig = gpu.merge_topk(D, I)                               # host pointers
    assert np.array_equal(Ig, Ir) and np.array_equal(Dg, Dr), f"ns={ns} k={k} ties={ties}: host arrays"
    Dt, It = gpu.merge_topk(torch.from_numpy(D).cuda(), torch.from_numpy(I).cuda())
    assert np.array_equal(It.cpu().numpy(), Ir) and np.array_equal(Dt.cpu().numpy(), Dr), f"ns={ns} k={k} ties={ties}: CUDA tensors"
    # the packed form of the RCCL exchange: one pack_topk block per shard (offset 0), gathered, merged on the current stream
    blocks = [gpu.pack_topk(torch.from_numpy(D[s]).cuda(), torch.from_numpy(I[s]).cuda(), 0) for s in range(ns)]
    Dp, Ip = gpu.merge_packed(torch.stack(blocks))
    assert np.array_equal(Ip.cpu().numpy(), Ir) and np.array_equal(Dp.cpu().numpy(), Dr), f"ns={ns} k={k} ties={ties}: packed"


def test_merge_refuses_k_above_8192(gpu):
    """... only where the merge needs rounds (more than 16384 keys per query): one shard of k = 16384 is a single launch."""
    D = np.zeros((2, 1, 8193), np.float32); I = np.zeros((2, 1, 8193), np.int64)
    with pytest.raises(RuntimeError):
        gpu.merge_topk(D, I)
    D = np.zeros((1, 1, 16385), np.float32); I = np.zeros((1, 1, 16385), np.int64)
    with pytest.raises(RuntimeError):
        gpu.merge_topk(D, I)
