"""The committed PMC stamp (profiles/pmc_traffic.json: HBM bytes per scan launch and the MFMA-busy fraction that bench.py prints as
roofline.traffic / roofline.mfma_busy) must describe THIS tree: bench.py drops a stamp whose source hash differs (traffic: null), and a
commit that changes a hashed kernel source without a new PMC pass would ship a bench line without its traffic figure."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_the_stamp_was_measured_on_these_sources():
    import bench
    t = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    assert t["source_sha256"] == bench.kernel_source_hash(), "re-run tools/gpu_round.sh pmc_fetch and copy gpurun_out/pmc_traffic.json to profiles/"
    traffic, note, busy = bench.load_pmc_traffic(t["n"], t["n_gpus"], t["kernel"])
    assert traffic == t["hbm_bytes_per_launch"] and "same sources" in note
    # plausibility: between the algorithmic bytes of the headline (96 B x 100M) and twice that; MFMA busy a fraction
    assert 9.6e9 <= traffic <= 19.2e9
    assert busy is None or 0.0 < busy < 1.0


def test_a_stamp_of_another_workload_or_kernel_is_refused():
    import bench
    t = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    assert bench.load_pmc_traffic(t["n"] + 1, t["n_gpus"], t["kernel"])[0] is None
    assert bench.load_pmc_traffic(t["n"], t["n_gpus"] + 1, t["kernel"])[0] is None
    assert bench.load_pmc_traffic(t["n"], t["n_gpus"], "k_other")[0] is None
