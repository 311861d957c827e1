"""Indexer — the facade the pipeline stages and the serving API construct
(mirror of reference src/indicies/base.py:14-77; used at src/index.py:56-57, src/search.py:293,
api/api_index.py:28-29).  File names, dispatch and the search() return are the reference's.
"""
import logging
import os

from src.indicies.flat import FlatIndexer
from src.indicies.index_utils import get_index_dir_and_embedding_paths
from src.indicies.ivf_flat import IVFFlatIndexer
from src.indicies.ivf_pq import IVFPQIndexer


def index_file_names(index_args):
    """(index file name, has a .trained sibling) for cfg.datastore.index — reference base.py:23-27."""
    index_type = index_args.index_type
    if "IVF" in index_type:
        name = f"index_{index_type}.{index_args.sample_train_size}.{index_args.projection_size}.{index_args.ncentroids}.faiss"
        return name, True
    return f"index_{index_type}.faiss", False


class Indexer(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.args = cfg.datastore.index
        self.index_type = self.args.index_type
        # optional key (absent in the reference's configs -> one GPU, unchanged behaviour): cfg.datastore.index.devices =
        # [0, 1, ...] or "all" makes every engine object of this Indexer ONE handle over those GPUs (rsx_sharded_create):
        # the single index.search(all_queries, k) call of src/search.py:296 then spans the node
        # Both selections are scoped to THIS constructor (every engine object of an Indexer is created here: read_index /
        # _new_index) and restored afterwards, so a later Indexer whose config omits the keys does not inherit them.
        devices = self._opt("devices", None)
        import rsx
        from src.indicies import engine as _engine_sel
        prev_devices, prev_backend = rsx.get_default_devices(), _engine_sel.backend_name()
        rsx.set_default_devices(None if devices is None else (devices if isinstance(devices, str) else list(devices)))
        # optional keys of SURVEY 8b, same rule (absent -> the defaults preserve behaviour): `backend` selects the engine
        # module ("mi355x" | "faiss"), `storage_dtype` ("auto" | "float16") is checked once the index exists
        _engine_sel.set_backend(self._opt("backend", "mi355x"))
        self.backend = _engine_sel.backend_name()
        try:
            self._construct(cfg, _engine_sel)
        finally:
            rsx.set_default_devices(prev_devices)
            _engine_sel.set_backend(prev_backend)

    def _construct(self, cfg, _engine_sel):
        passage_dir = self.cfg.datastore.embedding.passages_dir
        index_dir, embedding_paths = get_index_dir_and_embedding_paths(cfg)
        os.makedirs(index_dir, exist_ok=True)
        logging.info(f"Indexing for passages: {embedding_paths}")
        name, has_trained = index_file_names(self.args)
        index_path = os.path.join(index_dir, name)
        meta_file = os.path.join(index_dir, name + ".meta")
        trained_index_path = os.path.join(index_dir, name + ".trained") if has_trained else None
        pos_map_save_path = os.path.join(index_dir, "passage_pos_id_map.pkl")

        common = dict(embed_paths=embedding_paths, index_path=index_path, meta_file=meta_file,
                      passage_dir=passage_dir, pos_map_save_path=pos_map_save_path,
                      dimension=self.args.projection_size)
        if self.index_type == "Flat":
            self.datastore = FlatIndexer(**common)
        elif self.index_type == "IVFFlat":
            self.datastore = IVFFlatIndexer(trained_index_path=trained_index_path,
                                            sample_train_size=self.args.sample_train_size, prev_index_path=None,
                                            ncentroids=self.args.ncentroids, probe=self.args.probe, **common)
        elif self.index_type == "IVFPQ":
            self.datastore = IVFPQIndexer(trained_index_path=trained_index_path,
                                          sample_train_size=self.args.sample_train_size, prev_index_path=None,
                                          ncentroids=self.args.ncentroids, probe=self.args.probe,
                                          n_subquantizers=self.args.n_subquantizers, code_size=self.args.n_bits,
                                          **common)
        else:
            raise NotImplementedError
        if self.index_type in ("Flat", "IVFFlat"):
            _engine_sel.check_storage_dtype(self.datastore.index, self._opt("storage_dtype", "auto"))

    def _opt(self, key, default):
        """optional cfg.datastore.index key (OmegaConf node, dict-like or attribute container)"""
        try:
            v = self.args.get(key, default) if hasattr(self.args, "get") else getattr(self.args, key, default)
        except (KeyError, AttributeError):
            v = default
        return default if v is None else v

    def search(self, query_embs, k=5):
        all_scores, all_passages, db_ids = self.datastore.search(query_embs, k)
        return all_scores, all_passages, db_ids
