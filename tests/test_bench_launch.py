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
