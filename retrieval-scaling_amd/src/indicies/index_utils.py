"""Artefact layout, embedding-shard I/O, passage position maps and the shared backend base.

Host-side mirror of the reference's src/indicies/index_utils.py (path and id plumbing, SURVEY §8
rows a5, a9, a10).  Same function names, arguments and on-disk names; the implementation is new.
`cfg` may be an OmegaConf node (reference) or any attribute/dict container (`cfg_get`).
"""
import glob
import collections
import json
import os
import pickle
import re

import numpy as np


def cfg_get(node, key, default=None):
    """OmegaConf-style `.get` that also works on plain dicts / attribute objects."""
    if node is None:
        return default
    if hasattr(node, "get") and not isinstance(node, (list, tuple)):
        try:
            v = node.get(key, default)
            return default if v is None else v
        except TypeError:
            pass
    return getattr(node, key, default)


def _shard_number(path, prefix):
    return int(os.path.basename(path).split(f"{prefix}_")[-1].split(".pkl")[0])


def get_index_dir_and_embedding_paths(cfg, index_shard_ids=None):
    """reference index_utils.py:9-34.  index dir = <embedding_dir>/index_<type>/<ids joined by _>
    when shard ids are given, else <dir of the globbed files>/index_<type>."""
    emb = cfg.datastore.embedding
    idx = cfg.datastore.index
    index_type = idx.index_type
    if index_shard_ids is None:
        index_shard_ids = cfg_get(idx, "index_shard_ids", None)
    index_shard_ids = sorted(index_shard_ids)  # invariant to the order given (raises on None, as the reference)
    if index_shard_ids:
        shard_ids = [int(i) for i in index_shard_ids]
        embedding_paths = [os.path.join(emb.embedding_dir, f"{emb.prefix}_{s:02d}.pkl") for s in shard_ids]
        name = "_".join(str(s) for s in sorted(shard_ids))
        index_dir = os.path.join(os.path.dirname(embedding_paths[0]), f"index_{index_type}/{name}")
    else:
        embedding_paths = sorted(glob.glob(idx.passages_embeddings), key=lambda p: _shard_number(p, emb.prefix))
        if idx.num_subsampled_embedding_files != -1:
            embedding_paths = embedding_paths[: idx.num_subsampled_embedding_files]
        index_dir = os.path.join(os.path.dirname(embedding_paths[0]), f"index_{index_type}")
    return index_dir, embedding_paths


def convert_pkl_to_jsonl(passage_dir):
    """reference index_utils.py:38-68: every passages *.pkl gets a sibling *.jsonl (one item per line)."""
    if os.path.isdir(passage_dir):
        files = [f for f in os.listdir(passage_dir) if ".pkl" in f and "pos_id_map" not in f]
    elif os.path.isfile(passage_dir):
        assert ".pkl" in passage_dir
        files = [passage_dir]
    else:
        raise AssertionError(f"{passage_dir} does not exist or is neither a file nor a directory.")
    for name in files:
        src = os.path.join(passage_dir, name)
        dst = src.replace(".pkl", ".jsonl")
        if os.path.exists(dst):
            continue
        with open(src, "rb") as f:
            items = pickle.load(f)
        with open(dst, "w") as f:
            for item in items:
                f.write(json.dumps(item))
                f.write("\n")


def _line_offsets(path):
    """{doc_id: [path, byte offset of line doc_id]} — offsets as text-mode tell() reports them."""
    table = {}
    with open(path, "rb") as f:
        pos, doc = 0, 0
        for line in f:
            table[doc] = [path, pos]
            pos += len(line)
            doc += 1
    return table


def get_passage_pos_ids(passage_dir, pos_map_save_path):
    """reference index_utils.py:71-134: {shard_id: {chunk_id: [file, offset]}}, cached as a pickle."""
    if pos_map_save_path is not None and os.path.exists(pos_map_save_path):
        with open(pos_map_save_path, "rb") as f:
            return pickle.load(f)
    pos_id_map = {}
    if os.path.isdir(passage_dir):
        for name in os.listdir(passage_dir):
            if ".jsonl" not in name or "pos_id_map" in name:
                continue
            m = re.match(r"raw_passages-(\d+)-of-\d+\.jsonl", name)
            pos_id_map[int(m.group(1))] = _line_offsets(os.path.join(passage_dir, name))
    elif os.path.isfile(passage_dir):
        jsonl = passage_dir.replace(".pkl", ".jsonl")
        assert ".pkl" in passage_dir and os.path.exists(jsonl)
        m = re.search(r"-(\d+)-of-\d+\.pkl", passage_dir)
        assert m, f"Cannot extract shard_id from {passage_dir}"
        pos_id_map[int(m.group(1))] = _line_offsets(jsonl)
    else:
        raise AssertionError(f"{passage_dir} does not exist or is neither a file nor a directory.")
    if pos_map_save_path is not None:
        with open(pos_map_save_path, "wb") as f:
            pickle.dump(pos_id_map, f)
    return pos_id_map


def shard_id_of(embed_path):
    """Absolute shard id encoded in `<prefix>_<NN>.pkl` (reference flat.py:54-55,77)."""
    return int(re.search(r"_(\d+)\.pkl$", embed_path).group(1))


def load_embedding_shard(embed_path):
    """Embedding shard = pickle (ids: list, embs: np.ndarray [n, d] fp16|fp32) — written by
    src/embed.py:155-156.  Returned WITHOUT the reference's fp32 up-cast (flat.py:86): fp16 values
    stay fp16 all the way into HBM, which is lossless."""
    with open(embed_path, "rb") as fin:
        ids, embeddings = pickle.load(fin)
    embeddings = np.asarray(embeddings)
    if embeddings.dtype not in (np.float16, np.float32):
        embeddings = embeddings.astype(np.float32)
    return ids, np.ascontiguousarray(embeddings)


class BackendBase:
    """What FlatIndexer / IVFFlatIndexer / IVFPQIndexer share in the reference (each file there
    carries its own copy): meta/position-map files, id -> passage lookup, the search return leg."""

    index = None
    index_id_to_db_id = None
    psg_pos_id_map = None

    # ---- files
    def load_index_id_to_db_id(self):
        with open(self.meta_file, "rb") as reader:
            return pickle.load(reader)

    def _save_meta(self):
        with open(self.meta_file, "wb") as fout:
            pickle.dump(self.index_id_to_db_id, fout)

    def build_passage_pos_id_map(self):
        convert_pkl_to_jsonl(self.passage_dir)
        return get_passage_pos_ids(self.passage_dir, self.pos_map_save_path)

    def load_psg_pos_id_map(self):
        if os.path.exists(self.pos_map_save_path):
            with open(self.pos_map_save_path, "rb") as f:
                return pickle.load(f)
        return self.build_passage_pos_id_map()

    # ---- embeddings
    def load_embeds(self, shard_id=None):
        chunks = []
        for embed_path in self.embed_paths:
            if shard_id is not None and shard_id_of(embed_path) != shard_id:
                continue
            chunks.append(load_embedding_shard(embed_path)[1])
        return np.concatenate(chunks, axis=0) if len(chunks) != 1 else chunks[0]

    def get_embs(self, indices=None, shard_id=None):
        if shard_id is not None:
            return self.load_embeds(shard_id)
        raise AttributeError("get_embs(indices=...) needs resident embeddings, which no backend keeps")

    # ---- id -> passage (reference flat.py:115-136).  The reference opens the passage file once per
    # retrieved id (nq * k open() calls per search — the wall-clock bottleneck once the search itself is
    # fast); here the files stay open in a small LRU and a line is fetched by seek +
    # readline on the cached handle.  Same bytes, same json, same return value.
    _MAX_OPEN_PASSAGE_FILES = 64

    def _passage_file(self, filename):
        cache = self.__dict__.setdefault("_psg_files", collections.OrderedDict())
        f = cache.get(filename)
        if f is None:
            f = open(filename, "rb")
            cache[filename] = f
            if len(cache) > self._MAX_OPEN_PASSAGE_FILES:
                cache.popitem(last=False)[1].close()
        else:
            cache.move_to_end(filename)
        return f

    def close_passage_files(self):
        for f in self.__dict__.pop("_psg_files", {}).values():
            f.close()

    def _id2psg(self, shard_id, chunk_id):
        filename, position = self.psg_pos_id_map[shard_id][chunk_id]
        f = self._passage_file(filename)
        f.seek(position)
        return json.loads(f.readline().decode("utf-8"))

    def _db_id(self, index_id):
        # NOTE reference quirk kept: index_id == -1 (fewer than k hits) indexes the LAST element.
        return self.index_id_to_db_id[index_id]

    def _get_passage(self, index_id):
        db_id = self._db_id(index_id)
        try:
            shard_id, chunk_id = db_id
        except (TypeError, ValueError):  # legacy metas hold a scalar chunk id (flat.py:123-126)
            shard_id, chunk_id = 0, db_id
        return self._id2psg(shard_id, chunk_id)

    def get_retrieved_passages(self, all_indices):
        passages, db_ids = [], []
        texts = {}                      # a passage retrieved for several queries of the batch is read once
        for query_indices in all_indices:
            row = []
            for i in query_indices:
                i = int(i)
                if i not in texts:
                    texts[i] = self._get_passage(i)["text"]
                row.append(texts[i])
            passages.append(row)
            db_ids.append([self._db_id(int(i)) for i in query_indices])
        return passages, db_ids

    # ---- search (reference flat.py:138-141; the reference's astype(np.float32) is dropped: the
    # engine takes fp16 queries as they come out of the encoder)
    def search(self, query_embs, k=4096):
        all_scores, all_indices = self.index.search(query_embs, k)
        all_passages, db_ids = self.get_retrieved_passages(all_indices)
        return all_scores.tolist(), all_passages, db_ids

    # ---- population shared by all backends: one embedding shard at a time, sequential ids
    def _add_shards(self, index):
        for n_done, embed_path in enumerate(self.embed_paths):
            shard_id = int(re.search(r"passages_(\d+)\.pkl", os.path.basename(embed_path)).group(1))
            to_add = self.get_embs(shard_id=shard_id)
            index.add(to_add)
            self.index_id_to_db_id.extend([[shard_id, chunk_id] for chunk_id in range(len(to_add))])
            print(f"Added {n_done + 1} / {len(self.embed_paths)} shards")
