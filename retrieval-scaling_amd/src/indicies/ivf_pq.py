"""IVFPQIndexer — IVF-PQ, inner product, by_residual (mirror of reference src/indicies/ivf_pq.py).

Identical life-cycle to IVFFlatIndexer plus the two PQ arguments of the reference constructor
(ivf_pq.py:37-54): n_subquantizers (M) and code_size (bits per code; the reference passes
cfg.datastore.index.n_bits here, base.py:68).
"""
import numpy as np

from src.indicies.engine import engine
from src.indicies.ivf_flat import IVFFlatIndexer


class IVFPQIndexer(IVFFlatIndexer):
    def __init__(self, embed_paths, index_path, meta_file, trained_index_path, passage_dir=None,
                 pos_map_save_path=None, sample_train_size=1000000, prev_index_path=None, dimension=768,
                 dtype=np.float16, ncentroids=4096, probe=2048, num_keys_to_add_at_a_time=1000000,
                 DSTORE_SIZE_BATCH=51200000, n_subquantizers=16, code_size=8):
        self.n_subquantizers = n_subquantizers
        self.code_size = code_size
        super().__init__(embed_paths, index_path, meta_file, trained_index_path, passage_dir=passage_dir,
                         pos_map_save_path=pos_map_save_path, sample_train_size=sample_train_size,
                         prev_index_path=prev_index_path, dimension=dimension, dtype=dtype, ncentroids=ncentroids,
                         probe=probe, num_keys_to_add_at_a_time=num_keys_to_add_at_a_time,
                         DSTORE_SIZE_BATCH=DSTORE_SIZE_BATCH)

    def _new_index(self):
        quantizer = engine().IndexFlatIP(self.dimension)
        return engine().IndexIVFPQ(quantizer, self.dimension, self.ncentroids, self.n_subquantizers, self.code_size,
                              engine().METRIC_INNER_PRODUCT)
