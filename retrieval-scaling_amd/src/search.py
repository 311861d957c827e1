"""Offline dense-search driver and multi-shard merge (mirror of reference src/search.py:126-183,
213-373,810-831 — SURVEY §8 rows a6, a7).

What is kept: one `Indexer(cfg).search(all_query_embeddings, n_docs)` call per index, the
`ctxs` record layout, output paths, skip-if-exists / overwrite, `safe_write_jsonl`, and the merge
semantics (running concat, stable sort by float(score) descending, cut to n_docs).
What is not here: query-encoder loading (stays on stock PyTorch-ROCm; pass `query_encoder_fn` or
pre-computed `questions_embedding`), multi-domain merge / MinHash dedup / BM25 (text
post-processing, out of scope).
"""
import copy
import json
import logging
import os
import pickle as pkl

import numpy as np

from src.index import _shard_id_groups
from src.indicies.base import Indexer
from src.indicies.index_utils import cfg_get


def add_passages_to_eval_data(data, passages, scores, db_ids, valid_query_idx, domain=None):
    """reference src/search.py:126-146: scores serialised with str(); examples without a query get [None]."""
    assert len(valid_query_idx) == len(passages)
    valid = set(valid_query_idx)
    idx = 0
    for i, ex in enumerate(data):
        if i not in valid:
            ex["ctxs"] = [None]
            continue
        n_ctx = len(passages[0])
        ex["ctxs"] = [
            {"id": db_ids[idx][c], "source": domain, "retrieval text": passages[idx][c],
             "retrieval score": str(scores[idx][c])}
            for c in range(n_ctx)
        ]
        idx += 1


def _results_name(cfg):
    return os.path.basename(cfg.evaluation.data.eval_data).replace(".jsonl", "_retrieved_results.jsonl")


def get_search_output_path(cfg, index_shard_ids):
    postfix = "_".join(str(s) for s in index_shard_ids)
    return os.path.join(cfg.evaluation.eval_output_dir, postfix, _results_name(cfg))


def get_merged_search_output_path(cfg):
    groups = sorted(_shard_id_groups(cfg.datastore.index), key=lambda g: int(g[0]))
    merged = "-".join("_".join(str(s) for s in g) for g in groups)
    return os.path.join(cfg.evaluation.eval_output_dir, merged, _results_name(cfg))


def safe_write_jsonl(data, output_file):
    """Write all or nothing (reference src/search.py:810-824)."""
    ok = False
    try:
        with open(output_file, "w") as fout:
            for ex in data:
                fout.write(json.dumps(ex) + "\n")
        ok = True
        logging.info(f"Saved results to {output_file}")
    except Exception as e:  # noqa: BLE001 — the reference swallows and cleans up
        print(f"An error occurred: {e}")
    finally:
        if not ok and os.path.exists(output_file):
            os.remove(output_file)
            print(f"File '{output_file}' has been deleted due to an error.")


def merge_ctxs(per_shard_ctxs, n_docs):
    """The reference's merge rule for one example (src/search.py:358-367): extend shard by shard,
    after each shard stable-sort by float(score) descending and keep n_docs."""
    merged = []
    for i, ctxs in enumerate(per_shard_ctxs):
        if i == 0:
            merged = list(ctxs)
            continue
        merged.extend(ctxs)
        if merged and merged[0] is not None:
            merged = sorted(merged, key=lambda x: float(x["retrieval score"]), reverse=True)[:n_docs]
    return merged


def post_hoc_merge_topk(cfg):
    """Merge the per-index JSONL results (reference src/search.py:312-373)."""
    output_path = get_merged_search_output_path(cfg)
    if os.path.exists(output_path) and not cfg.evaluation.search.overwrite:
        print(f"The merged path exists, skipping...\n{output_path}")
        return None
    groups = _shard_id_groups(cfg.datastore.index)
    if len(groups) <= 1:
        print("Single-index mode: no need to merge")
        return None
    n_docs = cfg.evaluation.search.n_docs
    merged_data = []
    for i, shard_ids in enumerate(groups):
        shard_data = []
        with open(get_search_output_path(cfg, shard_ids), "r") as f:
            for line in f:
                try:
                    ex = json.loads(line)
                except json.JSONDecodeError:
                    continue
                if not ex["ctxs"] or ex["ctxs"][0] is None:
                    ex["ctxs"] = []
                shard_data.append(ex)
        if i == 0:
            merged_data = shard_data
            continue
        for cur, new in zip(merged_data, shard_data):
            assert cur["raw_query"] == new["raw_query"]
            cur["ctxs"] = merge_ctxs([cur["ctxs"], new["ctxs"]], n_docs)
            if cur["ctxs"]:
                assert len(cur["ctxs"]) == n_docs
    os.makedirs(os.path.dirname(output_path), exist_ok=True)
    safe_write_jsonl(merged_data, output_path)
    return output_path


def search_dense_topk(cfg, data=None, questions_embedding=None, query_encoder_fn=None):
    """reference src/search.py:213-309.

    data: list of eval examples with "raw_query" (reference: load_eval_data(cfg)).
    questions_embedding: [n_valid_queries, d] array, or None to call query_encoder_fn(queries)
    (the reference loads a HF encoder here; that stage is unchanged PyTorch and is injected).
    """
    eval_args = cfg.evaluation
    groups = _shard_id_groups(cfg.datastore.index)
    all_exist = all(os.path.exists(get_search_output_path(cfg, g)) for g in groups)
    if all_exist and not eval_args.search.overwrite:
        logging.info("All search results exist, skipping searching.")
    else:
        assert data is not None, "search_dense_topk needs the evaluation examples"
        queries, valid_query_idx = [], []
        for idx, ex in enumerate(data):
            if ex["raw_query"]:
                queries.append(ex["raw_query"])
                valid_query_idx.append(idx)
        cache = cfg_get(eval_args.search, "query_embedding_save_path", "")
        if questions_embedding is None and cfg_get(eval_args.search, "cache_query_embedding", False) and os.path.exists(cache):
            with open(cache, "rb") as fin:
                questions_embedding = pkl.load(fin)
        if questions_embedding is None:
            assert query_encoder_fn is not None, "no query embeddings and no encoder given"
            questions_embedding = query_encoder_fn(queries)
        if cfg_get(eval_args.search, "cache_query_embedding_only", False):
            return
        if not hasattr(questions_embedding, "is_cuda"):      # a CUDA tensor from the encoder goes to the engine as it is
            questions_embedding = np.asarray(questions_embedding)
        for shard_ids in groups:
            output_path = get_search_output_path(cfg, shard_ids)
            if os.path.exists(output_path) and not eval_args.search.overwrite:
                logging.info(f"{output_path} exists, skipping searching.")
                continue
            copied = copy.deepcopy(data)
            index = Indexer(cfg)
            # ONE call with every query, exactly like the reference (:296)
            all_scores, all_passages, db_ids = index.search(questions_embedding, eval_args.search.n_docs)
            add_passages_to_eval_data(copied, all_passages, all_scores, db_ids, valid_query_idx,
                                      domain=cfg_get(cfg.datastore, "domain", None))
            os.makedirs(os.path.dirname(output_path), exist_ok=True)
            safe_write_jsonl(copied, output_path)
    if cfg_get(eval_args.search, "merge_multi_index_results", True):
        post_hoc_merge_topk(cfg)


def search_topk(cfg, **kw):
    if cfg_get(cfg.model, "sparse_retriever", None):
        raise NotImplementedError("BM25 search is outside the dense-retrieval path")
    return search_dense_topk(cfg, **kw)
