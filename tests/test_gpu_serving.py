"""GPU: the serving twin (reference api/api_index.py:21-70, 88-95) over librsx — DatastoreAPI(cfg, shard_id).search(query,
n_docs) -> {'scores', 'passages', 'IDs'} and the 30-call latency protocol, with the query encoder injected (a lookup into the
shard embeddings; the encoder itself stays on PyTorch-ROCm and is out of scope).  One GPU handle, and ONE handle over two
shards (cfg.datastore.index.devices) — the in-node replacement of the HTTP fan-out of api/serve_main_node.py:281-323."""
import os

import numpy as np
import pytest

from test_host_logic import make_cfg, write_datastore

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("index_type", ["Flat", "IVFFlat", "IVFPQ"])
@pytest.mark.parametrize("two_shards", [False, True])
def test_datastore_api_on_gpu(gpu, orc, tmp_path, index_type, two_shards):
    from api.api_index import DatastoreAPI, get_datastore, profile_time
    tmp = str(tmp_path)
    embs = write_datastore(tmp, orc)
    table = {"q five": embs[0][5], "q seven": embs[1][7]}

    def enc(queries):
        return np.stack([table[s] for s in queries], 0)

    cfg = make_cfg(tmp, index_type, [0, 1], ncentroids=4, probe=4, n_subquantizers=8, sample_train_size=300)
    if two_shards:
        n = gpu.get_num_gpus()
        cfg.datastore.index["devices"] = [0, 1 % n]
    ds = get_datastore(cfg, query_encoder_fn=enc)
    assert ds.index.index.nshards == (2 if two_shards else 0)          # the engine object behind the backend
    assert gpu.get_default_devices() is None                            # the device list was scoped to that Indexer
    r = ds.search("q five", 3)
    assert set(r) == {"scores", "passages", "IDs"} and len(r["scores"]) == 1 and len(r["IDs"][0]) == 3
    r2 = ds.search(["q five", "q seven"], 3)
    allx = np.concatenate(embs, 0).astype(np.float32)
    qs = np.stack([table["q five"], table["q seven"]]).astype(np.float32)
    if index_type != "IVFPQ":      # probe == ncentroids: exhaustive and exact -> brute force over both shards
        D, I = orc.flat_search(qs, allx, 3, 0)
        assert r["IDs"] == [[[int(i) // 400, int(i) % 400] for i in I[0]]] and r["scores"] == [D[0].tolist()]
        assert r2["scores"] == D.tolist()
        assert r2["passages"][1][0] == f"shard {I[1, 0] // 400} chunk {I[1, 0] % 400} é"
    else:                          # PQ scores are approximate (8 sub-quantisers of 4 dims): structure + consistency
        for row_s, row_p, row_i in zip(r2["scores"], r2["passages"], r2["IDs"]):
            assert row_s == sorted(row_s, reverse=True) and len(row_i) == 3
            assert all(p == f"shard {s} chunk {c} é" for p, (s, c) in zip(row_p, row_i))
        assert r["scores"][0] == r2["scores"][0] and r["IDs"][0] == r2["IDs"][0]      # batch of 1 == row of a batch of 2
    with pytest.raises(AttributeError):
        ds.search(5)
    # one worker = one shard (api_index.py:23-27)
    one = DatastoreAPI(make_cfg(tmp, index_type, [0, 1], ncentroids=4, probe=4, n_subquantizers=8, sample_train_size=300),
                       shard_id=1, query_encoder_fn=enc)
    assert one.index.index.ntotal == 400
    got = one.search("q seven", 1)["IDs"]
    assert got[0][0][0] == 1 and (index_type == "IVFPQ" or got == [[[1, 7]]])       # ids are relative to that shard only
    per_query = profile_time(ds, "q five", 3)                            # 30 calls, 10 warm-up (api_index.py:88-95)
    assert 0.0 < per_query < 0.5
    with pytest.raises(AssertionError):
        profile_time(ds, "q five", 3, calls=5, warmup=10)
