"""Index-build stage entry (mirror of the dense part of reference src/index.py:46-57,205-209).

BM25 / pyserini (reference :82-202) is sparse Lucene retrieval — out of scope for the dense path.
"""
from src.indicies.base import Indexer
from src.indicies.index_utils import cfg_get


def _shard_id_groups(index_args):
    """Reference "multi-index mode": index_shard_ids may be a list of lists (src/index.py:49-54)."""
    ids = index_args.index_shard_ids
    first = ids[0]
    if isinstance(first, (list, tuple)) or type(first).__name__ == "ListConfig":
        return list(ids)
    return [ids]


def build_dense_index(cfg):
    index_args = cfg.datastore.index
    groups = _shard_id_groups(index_args)
    built = []
    for _ in groups:
        # Constructing the Indexer builds (or loads) the index as a side effect — as the reference
        # does (src/index.py:56-57, where the loop variable is likewise unused).
        built.append(Indexer(cfg))
    return built


def build_index(cfg):
    if cfg_get(cfg.model, "sparse_retriever", None):
        raise NotImplementedError("BM25 (pyserini) indexing is outside the dense-retrieval path")
    return build_dense_index(cfg)
