"""Multi-GPU search: vectors partitioned across ranks, one all-gather of per-shard top-k, merge.

The reference shards by running independent indexes (Slurm array / HTTP workers) and merging their
results by score in Python (src/search.py:312-373, api/serve_main_node.py:109-165).  Here the shards
are the GPUs of one node: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI),
every rank searches its own shard with the same query batch, ONE collective moves the packed
[nq, k] (score, id) candidates, and every rank runs the merge kernel (rsx_merge_topk) — result
order = the reference's: score descending, ties earlier shard first, then within-shard order.

The data path has no other collective: the scan itself never crosses GPUs.
"""
import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous id range [lo, hi) of `rank` — the reference's shard semantics (global ids are offsets)."""
    per = (n_total + world_size - 1) // world_size
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def merge_topk_host(D, I, metric=0):
    """Host restatement of the reference merge (src/search.py:362-367) on arrays [nshards, nq, k]:
    concat in shard order, stable sort by score (descending for IP), keep k; ids < 0 are padding.
    Used for JSONL-level merges and for CPU-tensor (gloo) gathers; GPUs use the rsx_merge_topk kernel."""
    D = np.asarray(D, dtype=np.float32)
    I = np.asarray(I, dtype=np.int64)
    ns, nq, k = D.shape
    Dc = np.transpose(D, (1, 0, 2)).reshape(nq, ns * k)
    Ic = np.transpose(I, (1, 0, 2)).reshape(nq, ns * k)
    Do = np.full((nq, k), -np.inf if metric == 0 else np.inf, dtype=np.float32)
    Io = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        valid = np.nonzero(Ic[q] >= 0)[0]
        key = -Dc[q, valid] if metric == 0 else Dc[q, valid]
        order = valid[np.argsort(key, kind="stable")][:k]
        Do[q, : len(order)] = Dc[q, order]
        Io[q, : len(order)] = Ic[q, order]
    return Do, Io


_SIGN = -(2 ** 63)


def raise_thresholds(taus):
    """Elementwise maximum of several threshold-key vectors (CUDA int64 views of UNSIGNED keys), written back to all of them:
    what the all-reduce(MAX) does across ranks, for several local handles (single-process tests / tools)."""
    import torch
    live = [t for t in taus if t is not None]
    if len(live) < 2:
        return
    m = live[0] ^ _SIGN
    for t in live[1:]:
        m = torch.maximum(m, t ^ _SIGN)
    m ^= _SIGN
    for t in live:
        t.copy_(m)


class ShardedSearcher:
    """Wraps a rank-local index whose vectors are ids [id_offset, id_offset + ntotal)."""

    def __init__(self, local_index, id_offset=0, group=None, metric=0, force_collective=False, exchange_thresholds=False):
        import torch.distributed as dist
        self.index = local_index
        self.id_offset = int(id_offset)
        self.group = group
        self.metric = metric
        self.force_collective = force_collective   # run the all-gather + merge even with one rank (tests)
        # LIST shards (rsx_set_param "add_list_mod"): a rank that does not own a query's closest lists derives a weak filter
        # threshold from its own lists; with exchange_thresholds the search runs in two calls and the ranks all-reduce(MAX)
        # the per-query threshold keys in between (one extra collective of nq x 8 bytes), so every rank filters as hard as
        # the single index would.  Vector shards see the same lists on every rank and do not need it.
        self.exchange_thresholds = exchange_thresholds
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # profile = True: CUDA events around the three device steps of the exchange (pack, collective, merge) on the caller's stream;
        # stage_ms() sums them per key (bench.py reports them beside the library's own stage times when world > 1)
        self.profile = False
        self._events = []

    def _mark(self, name, dev):
        if self.profile:
            import torch
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(dev))
            self._events.append((name, e))

    def stage_ms(self, reset=True):
        """{"pack": ms, "collective": ms, "merge": ms, "calls": n} accumulated since the last reset (synchronises the device)."""
        import torch
        out = {"pack": 0.0, "collective": 0.0, "merge": 0.0, "calls": 0}
        if self._events:
            torch.cuda.synchronize()
            for (n0, e0), (n1, e1) in zip(self._events[:-1], self._events[1:]):
                if n1 == "start":
                    continue
                out[n1] += e0.elapsed_time(e1)
            out["calls"] = sum(1 for n, _ in self._events if n == "start")
        if reset:
            self._events = []
        return out

    def _search_two_call(self, q, k):
        """One all-reduce(MAX) of the per-query threshold keys between rsx_search_prepass and rsx_search_scan.  The library runs
        ONE internal batch per two-call search, so a larger batch is cut into query_batch-sized pieces (every rank holds the
        same batch and the same knob, hence issues the same sequence of collectives).  Whatever happens between the two calls
        — a collective that times out, an allocation that fails — the parked search is always released with search_scan()
        before the error propagates: a handle must never be left with an open two-call search."""
        import torch
        import torch.distributed as dist
        qb = int(self.index._get("query_batch"))
        if q.shape[0] > qb:
            parts = [self._search_two_call(q[i:i + qb], k) for i in range(0, q.shape[0], qb)]
            return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
        tau = self.index.search_prepass(q, k)
        try:
            # every rank takes part in the collective, also one whose search has no pre-pass (it contributes "no threshold")
            t = (tau ^ _SIGN) if tau is not None else torch.full((q.shape[0],), _SIGN, dtype=torch.int64, device=q.device)
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            else:                                   # gloo (several ranks on one GPU, CPU tests): through the host
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t = h.to(q.device)
            if tau is not None:
                tau.copy_(t ^ _SIGN)
                torch.cuda.current_stream(q.device).synchronize()     # the library's stream reads the keys next
        except BaseException:
            try:
                self.index.search_scan()            # release the parked worker; its result is discarded
            except Exception:
                pass
            raise
        return self.index.search_scan()

    def search(self, q, k):
        """q: [nq, d] — the SAME batch on every rank (numpy / CPU tensor / CUDA tensor).
        Returns merged (D, I) [nq, k] on every rank, in the container type of the local result."""
        import torch
        import torch.distributed as dist
        if self.exchange_thresholds and torch.is_tensor(q) and q.is_cuda and self.world_size > 1:
            D, I = self._search_two_call(q, k)
        else:
            D, I = self.index.search(q, k)
        as_numpy = not torch.is_tensor(D)
        if as_numpy:
            D, I = torch.from_numpy(np.ascontiguousarray(D)), torch.from_numpy(np.ascontiguousarray(I))
        if D.is_cuda and not as_numpy and (self.world_size > 1 or self.force_collective):
            # GPU results: one pack kernel (adds the shard's id offset), one collective, one merge kernel,
            # all on the current stream
            import rsx
            nq = D.shape[0]
            self._mark("start", D.device)
            packed = rsx.pack_topk(D, I, self.id_offset)
            self._mark("pack", D.device)
            gathered = torch.empty((self.world_size, 2, nq, k), dtype=torch.int64, device=D.device)
            if dist.get_backend(self.group) == "nccl":
                dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), group=self.group)
            else:                                   # gloo (several ranks sharing one GPU: RCCL refuses that): the same block through the host
                hg = torch.empty((self.world_size, 2, nq, k), dtype=torch.int64)
                dist.all_gather_into_tensor(hg.view(-1), packed.cpu().view(-1), group=self.group)
                gathered.copy_(hg)
            self._mark("collective", D.device)
            out = rsx.merge_packed(gathered, metric=self.metric)      # any world_size x k <= 8192: rounds inside the library
            self._mark("merge", D.device)
            return out
        I = torch.where(I >= 0, I + self.id_offset, I)
        if self.world_size == 1 and not self.force_collective:
            return (D.numpy(), I.numpy()) if as_numpy else (D, I)
        nq = D.shape[0]
        # host results (or a gloo group): the same exchange with torch ops — one packed buffer -> one collective: [2, nq, k] int64 (scores as raw bits in the low word)
        backend = dist.get_backend(self.group)
        dev = D.device
        if backend == "nccl" and not D.is_cuda:
            dev = torch.device("cuda", torch.cuda.current_device())
        packed = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
        packed[0] = D.to(dev).contiguous().view(torch.int32).to(torch.int64)
        packed[1] = I.to(dev)
        gathered = torch.empty((self.world_size, 2, nq, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), group=self.group)
        Dall = gathered[:, 0].to(torch.int32).view(torch.float32).contiguous()
        Iall = gathered[:, 1].contiguous()
        if Dall.is_cuda:
            import rsx
            Dm, Im = rsx.merge_topk(Dall, Iall, metric=self.metric)
            if as_numpy:
                return Dm.cpu().numpy(), Im.cpu().numpy()
            return Dm, Im
        Dm, Im = merge_topk_host(Dall.numpy(), Iall.numpy(), self.metric)
        return (Dm, Im) if as_numpy else (torch.from_numpy(Dm), torch.from_numpy(Im))
