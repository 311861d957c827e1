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

    # ---- id -> passage (reference flat.py:115-136).  The reference resolves every retrieved id through two Python
    # containers (index_id_to_db_id list, {shard: {chunk: [file, offset]}} dict of dicts) and opens the passage file once
    # per id: nq * k open() + seek + readline calls per search — the wall-clock bottleneck once the search itself takes
    # milliseconds.  Here the two containers are flattened ONCE into integer arrays
    #     index id -> (file number, byte offset, line length)
    # and a batch is served by numpy indexing over its UNIQUE ids plus one os.pread per passage on a cached descriptor
    # (no seek state, no text-mode decoding layer, each passage read once per batch).  Same bytes, same json, same return
    # value.  The per-id helpers below (_id2psg, _get_passage) keep the reference's names for single look-ups.
    #
    # The -1 quirk (reference flat.py:124,133): an unfilled result slot carries id -1, which Python's negative indexing
    # maps to the LAST stored passage.  It is kept on purpose — the driver asserts exactly n_docs contexts per example
    # (src/search.py:367), so dropping the slot would change control flow — and it is explicit here: _row_of(-1) = n - 1.
    _MAX_OPEN_PASSAGE_FILES = 64

    def _passage_table(self):
        """(file names, file_no[idx], offset[idx], length[idx]) over index ids; built on first use, rebuilt when the id map
        grows (index still being populated)."""
        n = len(self.index_id_to_db_id)
        tab = self.__dict__.get("_psg_table")
        if tab is not None and tab[4] == n:
            return tab
        files, file_no_of = [], {}
        per_shard = {}                              # shard -> (file_no[chunk], offset[chunk]) dense over chunk ids
        for shard_id, chunks in self.psg_pos_id_map.items():
            m = (max(chunks) + 1) if chunks else 0
            fno = np.full(m, -1, dtype=np.int32); off = np.zeros(m, dtype=np.int64)
            for chunk_id, (filename, position) in chunks.items():
                k = file_no_of.get(filename)
                if k is None:
                    k = file_no_of[filename] = len(files)
                    files.append(filename)
                fno[chunk_id] = k; off[chunk_id] = position
            per_shard[shard_id] = (fno, off)
        # line length = distance to the next line start of the same file (the last line runs to the end of the file)
        ends = {}
        for fno, off in per_shard.values():
            for k in np.unique(fno[fno >= 0]):
                ends.setdefault(int(k), []).append(off[fno == k])
        nxt = {}
        for k, parts in ends.items():
            starts = np.unique(np.concatenate(parts))
            nxt[k] = (starts, np.append(starts[1:], os.path.getsize(files[k])))
        file_no = np.full(n, -1, dtype=np.int32); offset = np.zeros(n, dtype=np.int64); length = np.zeros(n, dtype=np.int64)
        if n:
            first = self.index_id_to_db_id[0]
            if isinstance(first, (list, tuple, np.ndarray)):
                ids = np.asarray(self.index_id_to_db_id, dtype=np.int64).reshape(n, 2)
                shard_of, chunk_of = ids[:, 0], ids[:, 1]
            else:                                   # legacy metas hold a scalar chunk id (flat.py:123-126): shard 0
                shard_of = np.zeros(n, dtype=np.int64); chunk_of = np.asarray(self.index_id_to_db_id, dtype=np.int64)
            for shard_id in np.unique(shard_of):
                sel = np.nonzero(shard_of == shard_id)[0]
                fno, off = per_shard[int(shard_id)]
                c = chunk_of[sel]
                file_no[sel] = fno[c]; offset[sel] = off[c]
            for k, (starts, stops) in nxt.items():
                sel = np.nonzero(file_no == k)[0]
                length[sel] = stops[np.searchsorted(starts, offset[sel])] - offset[sel]
        tab = (files, file_no, offset, length, n)
        self.__dict__["_psg_table"] = tab
        return tab

    def prepare_passage_table(self):
        """Build the id -> (file, offset, length) table NOW (the backends call this at the end of construction / load): for a
        100M-entry id map it is minutes of work and GBs of temporaries that must not land inside the first served query."""
        if self.psg_pos_id_map is not None and self.index_id_to_db_id is not None and len(self.index_id_to_db_id):
            self._passage_table()

    def _passage_fd(self, file_no, files):
        cache = self.__dict__.setdefault("_psg_fds", collections.OrderedDict())
        fd = cache.get(file_no)
        if fd is None:
            fd = os.open(files[file_no], os.O_RDONLY)
            cache[file_no] = fd
            if len(cache) > self._MAX_OPEN_PASSAGE_FILES:
                os.close(cache.popitem(last=False)[1])
        else:
            cache.move_to_end(file_no)
        return fd

    def close_passage_files(self):
        for fd in self.__dict__.pop("_psg_fds", {}).values():
            os.close(fd)
        for f in self.__dict__.pop("_psg_files", {}).values():
            f.close()

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
        try:
            idx = np.asarray(all_indices, dtype=np.int64)
        except ValueError:                                     # ragged rows
            idx = np.zeros(0, dtype=np.int64)
        if idx.ndim != 2 or self.psg_pos_id_map is None:      # ragged input / no position map: the per-id path
            passages = [[self._get_passage(int(i))["text"] for i in row] for row in all_indices]
            return passages, [[self._db_id(int(i)) for i in row] for row in all_indices]
        files, file_no, offset, length, n = self._passage_table()
        rows = np.where(idx < 0, idx + n, idx)                 # Python's negative indexing, made explicit (see above)
        uniq, inv = np.unique(rows, return_inverse=True)
        missing = uniq[file_no[uniq] < 0]
        if len(missing):                                       # the reference's dict lookup raises KeyError here (flat.py:131)
            raise KeyError(f"passage position map has no entry for db id {self.index_id_to_db_id[int(missing[0])]}")
        order = np.lexsort((offset[uniq], file_no[uniq]))      # file by file, front to back
        texts = [None] * len(uniq)
        for u in order:
            r = uniq[u]
            raw = os.pread(self._passage_fd(int(file_no[r]), files), int(length[r]), int(offset[r]))
            texts[u] = json.loads(raw.decode("utf-8"))["text"]
        inv = inv.reshape(idx.shape)
        passages = [[texts[j] for j in row] for row in inv]
        id_map = self.index_id_to_db_id
        db_ids = [[id_map[int(i)] for i in row] for row in idx]
        return passages, db_ids

    # ---- search (reference flat.py:138-141; the reference's astype(np.float32) is dropped: the
    # engine takes fp16 queries as they come out of the encoder)
    def search(self, query_embs, k=4096):
        """query_embs: numpy array (the reference) or a CUDA tensor straight from the encoder (no host round trip of the
        queries; D / I come back as tensors and only they cross PCIe)."""
        all_scores, all_indices = self.index.search(query_embs, k)
        if hasattr(all_indices, "is_cuda"):
            all_scores, all_indices = all_scores.cpu().numpy(), all_indices.cpu().numpy()
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
