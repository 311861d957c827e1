"""GPU: the WHOLE bench.py with N > 1 ranks, as a dry run on one GPU (task 3).

`bench.py --gpus N` had only ever executed with world_size 1: the parameter broadcast, the list-shard add stream, the
ShardedSearcher exchange inside the timed loop, the ground-truth all-gather, the elapsed all-reduce and the self-launch under
torch.distributed.run were dead code as far as any run was concerned.  Here N = 2 and N = 8 ranks share cuda:0
(`--share-gpu`) and exchange over gloo (`--dist-backend gloo`: RCCL refuses two ranks per device; the collectives are the
same calls routed through host memory), for both partitions of the index (`--shard vectors`: the reference's id-range
shards, src/search.py:282-303; `--shard lists`).  Every run must print ONE JSON line from rank 0 with n_gpus == N, the N = 1
run's recall@10, and the N = 1 run's ids and scores of the first timed batch (sha256 over D and I: the merged result of a
sharded index is the single index's, bit for bit — src/search.py:362-367 is the merge it replaces)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--n", "1600000", "--nlist", "256", "--chunk", "200000", "--steps", "2", "--warmup", "1", "--cpu-queries", "0",
          "--no-configs", "--no-faiss"]


def _run(extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + COMMON + extra, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, f"bench.py {extra} failed:\n{p.stderr[-3000:]}"
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def single(gpu):
    r = _run([])
    assert r["n_gpus"] == 1 and r["recall_at_10"] is not None
    return r


@pytest.mark.parametrize("ngpu,shard", [(2, "vectors"), (2, "lists"), (8, "vectors"), (8, "lists")])
def test_bench_py_with_n_ranks_on_one_gpu(gpu, single, ngpu, shard):
    r = _run(["--gpus", str(ngpu), "--share-gpu", "--dist-backend", "gloo", "--shard", shard])
    assert r["n_gpus"] == ngpu and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["dist_backend"] == "gloo"
    assert r["value"] > 0 and r["ms_per_step"] > 0
    assert r["scaling"] == "strong"
    assert r["recall_at_10"] == single["recall_at_10"], (r["recall_at_10"], single["recall_at_10"])
    assert r["recall_low_noise_queries"] == single["recall_low_noise_queries"]
    assert r["first_timed_batch_sha256"] == single["first_timed_batch_sha256"], "merged D / I of the sharded run differ from the single index"
    if shard == "lists":
        assert "inverted lists" in r["config"]["parallelism"]
    else:
        assert r["config"]["vectors_per_gpu"] == 1600000 // ngpu
    st = r["stage_ms_per_step"]      # round 5: the exchange is reported beside the library's stages when world > 1
    assert all(k_ in st for k_ in ("exchange_pack", "exchange_collective", "exchange_merge")), st
    assert st["exchange_collective"] > 0


def test_bench_py_weak_scaling_switch(gpu):
    """--scaling weak: --n is the shard of EVERY rank (fixed N/GPU), the line says so, and the merged result is
    that of a single index of N x n vectors."""
    r2 = _run(["--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--scaling", "weak", "--n", "800000"])
    assert r2["n_gpus"] == 2 and r2["scaling"] == "weak"
    assert r2["config"]["vectors_per_gpu"] == 800000 and r2["config"]["workload"].startswith("1600000x")
    r1 = _run([])                                     # the 1.6M-vector single index of COMMON
    assert r2["first_timed_batch_sha256"] == r1["first_timed_batch_sha256"]


def test_eight_ranks_at_the_reference_default_k(gpu):
    """k = 4096 (the reference backends' default, src/indicies/flat.py:138) on 8 ranks: 32768 keys per query reach the merge, which
    now runs in rounds (round 4 refused nshards * k > 16384).  Same ids and scores as the single index."""
    extra = ["--k", "4096", "--batch", "64", "--no-recall"]
    r1 = _run(extra)
    r8 = _run(extra + ["--gpus", "8", "--share-gpu", "--dist-backend", "gloo"])
    assert r8["n_gpus"] == 8 and r8["first_timed_batch_sha256"] == r1["first_timed_batch_sha256"]
