"""DatastoreAPI — the per-shard search object of the reference's serving tier (mirror of api/api_index.py:21-70).

The reference's workers (`api/serve_worker_node.py`) build one of these per datastore shard and call
`ds.search(query, n_docs)`; the main node fans out over HTTP and re-sorts (`api/serve_main_node.py:109-165, 281-323`).
Here the same object sits on the MI355X engine: `Indexer(cfg)` underneath (one GPU, or the whole node through
`cfg.datastore.index.devices`), so inside one node the HTTP fan-out is unnecessary — see INTEGRATION.md B' / C.

Out of scope, exactly as for `src/search.py`: loading the query encoder (Contriever / sentence-transformers / GritLM stay on
stock PyTorch-ROCm).  It is injected: `query_encoder_fn(list[str]) -> [n, d]` array or CUDA tensor.  The Flask / Slurm
plumbing around this class is control plane and is not mirrored.
"""
import time

from src.indicies.base import Indexer


class DatastoreAPI(object):
    def __init__(self, cfg, shard_id=None, query_encoder_fn=None):
        # api/api_index.py:23-27: a worker serves one shard id, or a list of them
        if shard_id is not None:
            cfg.datastore.index.index_shard_ids = shard_id if isinstance(shard_id, list) else [shard_id]
        self._index = Indexer(cfg)
        self.index = self._index.datastore
        self.query_encoder_fn = query_encoder_fn
        self.cfg = cfg

    def search(self, query, n_docs=3):
        """query: str or list[str] -> {'scores', 'passages', 'IDs'} (api/api_index.py:54-58)."""
        query_embedding = self.embed_query(query)
        searched_scores, searched_passages, db_ids = self.index.search(query_embedding, n_docs)
        return {"scores": searched_scores, "passages": searched_passages, "IDs": db_ids}

    def embed_query(self, query):
        if isinstance(query, str):
            queries = [query]
        elif isinstance(query, list):
            queries = query
        else:
            raise AttributeError("Query is not a string nor list!")       # the reference's error (api_index.py:66)
        if self.query_encoder_fn is None:
            raise RuntimeError("DatastoreAPI needs query_encoder_fn: the query encoder is not part of the search path")
        return self.query_encoder_fn(queries)


def get_datastore(cfg, shard_id=None, query_encoder_fn=None):
    return DatastoreAPI(cfg=cfg, shard_id=shard_id, query_encoder_fn=query_encoder_fn)


def profile_time(ds, query="Sunny San Diego days", n_docs=3, calls=30, warmup=10):
    """The reference's latency protocol (api/api_index.py:88-95): `calls` single-query searches, the first `warmup` are not
    timed; returns seconds per query.  (tools/bench_configs.py latency applies it to the engine alone.)"""
    assert calls > warmup >= 0, "profile_time: calls must exceed warmup (the reference protocol is 30 calls, 10 of them warm-up)"
    start = None
    for i in range(calls):
        if i == warmup:
            start = time.time()
        ds.search(query, n_docs)
    return (time.time() - start) / (calls - warmup)
