"""Offline dense-search driver and multi-shard merge (mirror of reference src/search.py:48-108, 126-183,
213-373, 810-831 — SURVEY §8 rows a6, a7).

What is kept: `search_topk(cfg)` exactly as `ric/main_ric.py:27-29` calls it, one
`Indexer(cfg).search(all_query_embeddings, n_docs)` call per index, the `ctxs` record layout, output paths,
skip-if-exists / overwrite, the query-embedding cache, `safe_write_jsonl`, and the merge semantics (running concat,
stable sort by float(score) descending, cut to n_docs).
The query encoder stays on stock PyTorch-ROCm: `load_query_encoder` / `embed_queries` follow the reference's dispatch on
the model name and import the host application's packages (`contriever`, `transformers`, `sentence_transformers`,
`gritlm`) and its `src.data.load_eval_data` only when they are needed — this file replaces the reference's
`src/search.py` inside the reference tree, where those modules live; outside it the missing module is named in the error.
`data=`, `questions_embedding=` and `query_encoder_fn=` remain as optional injections (tests, serving, other encoders).
What is not here: multi-domain merge / MinHash dedup / BM25 (text post-processing, out of scope).
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


def _host_module(name, why):
    """Import a module of the host application (the reference tree) on first use; name it clearly when it is absent."""
    import importlib
    try:
        return importlib.import_module(name)
    except ImportError as e:
        raise ImportError(f"search_topk(cfg) needs `{name}` ({why}); it is part of the host application "
                          f"(RulinShao/retrieval-scaling) and is imported only when no `data=` / `query_encoder_fn=` / "
                          f"`questions_embedding=` is passed: {e}") from e


def _device():
    import torch
    return "cuda" if torch.cuda.is_available() else "cpu"


def load_query_encoder(cfg):
    """(encoder, tokenizer, model_name) for cfg.model.query_encoder — the reference's dispatch (src/search.py:236-262):
    contriever / dragon, drama (HF AutoModel) / sentence-transformers, e5, Qwen3 / ReasonIR, GRIT; eval mode, on the GPU,
    half precision unless cfg.datastore.index.no_fp16."""
    name = cfg.model.query_encoder
    tok_name = cfg_get(cfg.model, "query_tokenizer", name)
    logging.info(f"Loading model from: {name}")
    tokenizer = None
    if "contriever" in name:
        encoder, tokenizer, _ = _host_module("contriever.src.contriever", "Contriever query encoder").load_retriever(name)
    elif "dragon" in name or "drama" in name:
        tf = _host_module("transformers", "HF query encoder")
        tokenizer = tf.AutoTokenizer.from_pretrained(tok_name)
        encoder = tf.AutoModel.from_pretrained(name, trust_remote_code=True)
    elif "sentence-transformers" in name or "e5" in name or "Qwen3" in name:
        encoder = _host_module("sentence_transformers", "SentenceTransformer query encoder").SentenceTransformer(name)
    elif "ReasonIR" in name or "GRIT" in name:
        encoder = _host_module("gritlm", "GritLM query encoder").GritLM(name, torch_dtype="auto", mode="embedding")
    else:
        print(f"{name} is not supported!")
        raise AttributeError(name)
    encoder.eval()
    encoder = encoder.to(_device())
    if not cfg_get(cfg.datastore.index, "no_fp16", False):
        encoder = encoder.half()
    return encoder, tokenizer, name


def embed_queries(args, queries, model, tokenizer, model_name_or_path):
    """reference src/search.py:48-108: lowercase / normalise, encode in batches of per_gpu_batch_size, return
    [n_queries, d] (numpy, as the reference; `args.keep_on_device` keeps the torch tensor in HBM for the engine —
    SURVEY §8f row 4), and write the cache file when args.cache_query_embedding."""
    import torch
    norm = None
    if cfg_get(args, "normalize_text", False):
        norm = _host_module("contriever.src.normalize_text", "normalize_text=true").normalize
    texts = []
    for q in queries:
        if cfg_get(args, "lowercase", False):
            q = q.lower()
        texts.append(norm(q) if norm else q)
    bs = int(cfg_get(args, "per_gpu_batch_size", 64))
    name = model_name_or_path
    if "sentence-transformers" in name or "e5" in name or "Qwen3" in name:
        kw = {"prompt_name": "query"} if "Qwen3" in name else {}
        embeddings = model.encode(texts, batch_size=min(128, bs), **kw)
    else:
        model.eval()
        dev = _device()
        outs = []
        with torch.no_grad():
            for b0 in range(0, len(texts), bs):
                batch = texts[b0:b0 + bs]
                if "drama" in name:
                    out = model.encode_queries(batch, batch, dim=768)
                elif "ReasonIR" in name or "GRIT" in name:
                    out = torch.as_tensor(model.encode(batch, instruction="", batch_size=bs))
                else:
                    enc = tokenizer.batch_encode_plus(batch, return_tensors="pt", padding=True, truncation=True,
                                                      max_length=cfg_get(args, "question_maxlength", 512))
                    out = model(**{k: v.to(dev) for k, v in enc.items()})
                    if "contriever" not in name:
                        out = out.last_hidden_state[:, 0, :]
                outs.append(out)
        embeddings = torch.cat([o.to(outs[0].device) for o in outs], dim=0) if outs else torch.zeros((0, 0))
        if not (cfg_get(args, "keep_on_device", False) and embeddings.is_cuda):
            embeddings = embeddings.cpu().numpy()
    print(f"Questions embeddings shape: {tuple(embeddings.shape)}")
    if cfg_get(args, "cache_query_embedding", False):
        with open(args.query_embedding_save_path, "wb") as fout:
            pkl.dump(embeddings.cpu().numpy() if hasattr(embeddings, "is_cuda") else embeddings, fout)
    return embeddings


def search_dense_topk(cfg, data=None, questions_embedding=None, query_encoder_fn=None):
    """reference src/search.py:213-309, callable with `cfg` alone (ric/main_ric.py:27-29).

    data: list of eval examples with "raw_query"; None -> the host application's `src.data.load_eval_data(cfg)` (:264).
    questions_embedding: [n_valid_queries, d] array; None -> the cache file (:276-279), else `query_encoder_fn(queries)`
    when given, else `load_query_encoder(cfg)` + `embed_queries(...)` (:236-281).
    """
    eval_args = cfg.evaluation
    groups = _shard_id_groups(cfg.datastore.index)
    all_exist = all(os.path.exists(get_search_output_path(cfg, g)) for g in groups)
    if all_exist and not eval_args.search.overwrite:
        logging.info("All search results exist, skipping searching.")
    else:
        if data is None:
            data = _host_module("src.data", "evaluation examples: load_eval_data(cfg)").load_eval_data(cfg)
        queries, valid_query_idx = [], []
        for idx, ex in enumerate(data):
            if ex["raw_query"]:
                queries.append(ex["raw_query"])
                valid_query_idx.append(idx)
        logging.info(f"Searching for {len(queries)} queries from {len(data)} total evaluation samples...")
        cache = cfg_get(eval_args.search, "query_embedding_save_path", "")
        if questions_embedding is None and cfg_get(eval_args.search, "cache_query_embedding", False) and os.path.exists(cache):
            logging.info(f"Loading query embeddings from {cache}")
            with open(cache, "rb") as fin:
                questions_embedding = pkl.load(fin)
        if questions_embedding is None and query_encoder_fn is not None:
            questions_embedding = query_encoder_fn(queries)
        if questions_embedding is None:
            encoder, tokenizer, name = load_query_encoder(cfg)
            questions_embedding = embed_queries(eval_args.search, queries, encoder, tokenizer, name)
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
    """reference src/search.py:827-831 — `search_topk(cfg)` is the whole call (ric/main_ric.py:29)."""
    if cfg_get(cfg.model, "sparse_retriever", None):
        raise NotImplementedError("BM25 search is outside the dense-retrieval path")
    return search_dense_topk(cfg, **kw)
