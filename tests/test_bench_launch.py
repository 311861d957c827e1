"""CPU: `python bench.py --gpus N` without a launcher must start N ranks by itself (it used to die on
`assert world == args.gpus`).  There is no GPU here, so each rank stops at the "needs a GPU" assertion — which proves the
ranks were started with WORLD_SIZE = N."""
import os
import subprocess
import sys

from conftest import REPO


def test_bench_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""          # also on a GPU box: this test is about the launch, not the bench
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "1000"],
                       env=env, capture_output=True, text=True, timeout=600)
    err = r.stderr
    assert r.returncode != 0
    assert "self-launch:" in err and "--nproc-per-node=2" in err
    assert "launch with torch.distributed.run" not in err            # the old failure
    assert err.count("bench.py needs a GPU") >= 2                     # both ranks got as far as the device check


def test_committed_bench_line_keeps_the_contract():
    """The newest full bench line under profiles/ (the driver's `python bench.py` on a GPU box, copied there by the evidence pass)
    carries every key of the bench contract, a roofline whose fraction is achieved / peak, and a CPU baseline that says what it
    is.  bench.py cannot run without a GPU: this guards the line's SHAPE between GPU sessions."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_ivfpq100M.json")), key=os.path.getmtime)
    assert files, "no committed bench line under profiles/"
    lines = [l for l in open(files[-1]) if l.startswith("{")]
    assert len(lines) == 1, "the bench prints ONE JSON line"
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["unit"] == "queries/s" and j["higher_is_better"] is True and j["n_gpus"] == 1 and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 1024 / (j["ms_per_step"] * 1e-3)) / j["value"] < 0.01, "value = queries of a step / time of a step (batch 1024)"
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # round 5: the MFMA-busy counter rides with the traffic stamp, and the same operating point is reported on four other distributions
    assert "mfma_busy" in r and "lds" in r
    od = j.get("other_distributions")
    assert od and set(od) >= {"hot_lists", "hot_probe_sets", "informative", "norm_skew"}, od and list(od)
    for name, leg in od.items():
        assert "error" not in leg, (name, leg.get("error"))
        assert leg["exact_fallback_queries_per_step"] == 0 and leg["oracle_parity_ids_and_scores"] is True, name
        assert leg["ms_per_step"] < 2 * j["ms_per_step"], (name, leg["ms_per_step"])       # no distribution slower than 2x the headline
    assert od["informative"]["recall_by_nprobe"]["nprobe32"]["recall_at_10"] > od["informative"]["recall_by_nprobe"]["nprobe1"]["recall_at_10"]
