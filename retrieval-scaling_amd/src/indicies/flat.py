"""FlatIndexer — exact inner-product index (mirror of reference src/indicies/flat.py).

Same constructor kwargs, attributes (.index, .index_id_to_db_id, .psg_pos_id_map, .cuda) and
search() return as the reference; the engine object is engine().IndexFlatIP (HBM-resident, fp16 rows,
MFMA scan + exact re-rank) instead of faiss.IndexFlatIP.
"""
import os
import time

from src.indicies.engine import engine
from src.indicies.index_utils import BackendBase


class FlatIndexer(BackendBase):
    def __init__(self, embed_paths=None, index_path=None, meta_file=None, passage_dir=None,
                 pos_map_save_path=None, dimension=768):
        self.embed_paths = embed_paths
        self.index_path = index_path
        self.meta_file = meta_file
        self.passage_dir = passage_dir
        self.pos_map_save_path = pos_map_save_path
        self.dimension = dimension
        self.cuda = True  # the index lives on the MI355X (reference: False, FAISS CPU)

        if os.path.exists(index_path) and os.path.exists(self.meta_file):
            print("Loading index...")
            self.index = engine().read_index(index_path)
            self.index_id_to_db_id = self.load_index_id_to_db_id()
        else:
            self.index = engine().IndexFlatIP(dimension)
            self.index_id_to_db_id = []
            print("Building index...")
            self._build_index()

        if self.pos_map_save_path is not None:
            self.psg_pos_id_map = self.load_psg_pos_id_map()
            self.prepare_passage_table()

    def _build_index(self):
        start_time = time.time()
        self._add_shards(self.index)
        engine().write_index(self.index, self.index_path)
        self._save_meta()
        print(f"Adding took {time.time() - start_time} s")
        print(f"Total data indexed {len(self.index_id_to_db_id)}")
