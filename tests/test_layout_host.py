"""CPU: the IVF-PQ code layouts (csrc/rsx_internal.h: pq_code_addr) checked on the host — one-to-one per slab, the bytes a scan
lane reads are the sub-quantisers its gather addresses assume, and every half wave addresses 32 different LDS banks at every step.
The check is a small host program compiled from the engine's own header with hipcc (no GPU, no kernel launch)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_pq_code_layouts_on_the_host(tmp_path):
    exe = str(tmp_path / "layout_check")
    src = os.path.join(REPO, "tests", "host", "layout_check.hip")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-value", "-o", exe, src])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "layouts ok" in out.stdout, out.stdout + out.stderr
