"""The C++ host mirror (rp-tree_amd/host/rptree.hpp) replays the reference's own integration
test (test/Data/RPTreeSpec.hs:50-85) through the C ABI, linked against the system HIP runtime
(no torch in the process)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rp-tree_amd", "host", "example_two_discs")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rp-tree_amd")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rp-tree_amd", "host")],
                          stdout=subprocess.DEVNULL)


def test_cpp_host_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "RPTError" in r.stdout      # no CPU fallback


@pytest.mark.gpu
def test_cpp_host_two_discs():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok:" in r.stdout
