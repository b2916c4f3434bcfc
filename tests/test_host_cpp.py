"""The C++ host layer (host/): classes shaped like the reference's tracer-facing C# classes on
top of the C ABI.  Its self-checking driver runs host-only checks on CPU and the full
render / per-pixel / per-ray checks on a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "test_host")


def _ensure_built():
    # make rebuilds only what is older than its sources (include/vrt.h and libvrt.so among them)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])


def test_host_layer_cpu_checks():
    _ensure_built()
    out = subprocess.run([BIN, "--cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_host: OK" in out.stdout


@pytest.mark.gpu
def test_host_layer_on_gpu():
    assert os.path.exists(BIN), "host/test_host must be built before the GPU run (__graft_entry__.build())"
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_host: OK" in out.stdout
