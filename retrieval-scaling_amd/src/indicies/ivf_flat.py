"""IVFFlatIndexer — IVF-Flat, inner product (mirror of reference src/indicies/ivf_flat.py).

sample -> train -> add -> write, `.trained` reuse and nprobe-at-load follow the reference
(ivf_flat.py:69-85,122-189); training and add run on the MI355X through rsx (the reference's
CUDA-only faiss.index_cpu_to_gpu branch, :152-163, is not needed).
"""
import os
import time

import numpy as np

from src.indicies.engine import engine
from src.indicies.index_utils import BackendBase, load_embedding_shard


class IVFFlatIndexer(BackendBase):
    def __init__(self, embed_paths, index_path, meta_file, trained_index_path, passage_dir=None,
                 pos_map_save_path=None, sample_train_size=1000000, prev_index_path=None, dimension=768,
                 dtype=np.float16, ncentroids=4096, probe=2048, num_keys_to_add_at_a_time=1000000,
                 DSTORE_SIZE_BATCH=51200000):
        self.embed_paths = embed_paths
        self.index_path = index_path
        self.meta_file = meta_file
        self.prev_index_path = prev_index_path
        self.trained_index_path = trained_index_path
        self.passage_dir = passage_dir
        self.pos_map_save_path = pos_map_save_path
        self.cuda = True

        self.sample_size = sample_train_size
        self.dimension = dimension
        self.ncentroids = ncentroids
        self.probe = probe
        self.num_keys_to_add_at_a_time = num_keys_to_add_at_a_time  # accepted, unused (as in the reference)

        if os.path.exists(index_path) and os.path.exists(self.meta_file):
            print("Loading index...")
            self.index = engine().read_index(index_path)
            self.index_id_to_db_id = self.load_index_id_to_db_id()
            self.index.nprobe = self.probe
        else:
            self.index_id_to_db_id = []
            if not os.path.exists(self.trained_index_path):
                print("Training index...")
                self._sample_and_train_index()
            print("Building index...")
            self.index = self._add_keys(self.index_path,
                                        self.prev_index_path if self.prev_index_path is not None else self.trained_index_path)

        if self.pos_map_save_path is not None:
            self.psg_pos_id_map = self.load_psg_pos_id_map()
            self.prepare_passage_table()

    # ---- engine object (overridden by IVFPQIndexer)
    def _new_index(self):
        quantizer = engine().IndexFlatIP(self.dimension)
        return engine().IndexIVFFlat(quantizer, self.dimension, self.ncentroids, engine().METRIC_INNER_PRODUCT)

    # ---- training (reference ivf_flat.py:122-167)
    def _sample_and_train_index(self):
        per_shard = self.sample_size // len(self.embed_paths)
        sampled = []
        for embed_path in self.embed_paths:
            _, embeddings = load_embedding_shard(embed_path)
            n = len(embeddings)
            pick = np.random.choice(np.arange(n), size=[min(per_shard, n)], replace=False)  # unseeded, as the reference
            sampled.append(embeddings[pick])
        sampled = np.concatenate(sampled, axis=0)
        start_time = time.time()
        self._train_index(sampled, self.trained_index_path)
        print("Finish training (%ds)" % (time.time() - start_time))

    def _train_index(self, sampled_embs, trained_index_path):
        start_index = self._new_index()
        start_index.nprobe = self.probe
        np.random.seed(1)
        start_index.train(sampled_embs)
        engine().write_index(start_index, trained_index_path)

    # ---- population (reference ivf_flat.py:169-189)
    def _add_keys(self, index_path, trained_index_path):
        index = engine().read_index(trained_index_path)
        assert index.is_trained and index.ntotal == 0
        start_time = time.time()
        self._add_shards(index)
        index.nprobe = self.probe
        engine().write_index(index, index_path)
        self._save_meta()
        print(f"Adding took {time.time() - start_time} s")
        return index
