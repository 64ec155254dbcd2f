"""The C ABI from a native host: tests/cabi/cabi_host_test.cpp links libbts_hip.so (and the oracle's C restatement as
the checker) with no Python or torch in the process -- LPG bit-exact vs the C oracle, a fused convolution vs host
loops, and the error codes for bad arguments.  Built by __graft_entry__.build() (tests/cabi/Makefile)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cabi", "_build", "cabi_host_test")


def test_native_host_calls_the_c_abi():
    # (re)build when the header, the source or the library changed; a box without hipcc uses the shipped binary
    mk = subprocess.run(["make", "-C", os.path.join(HERE, "cabi")], capture_output=True, text=True)
    assert mk.returncode == 0 or os.path.exists(BIN), mk.stdout + mk.stderr
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C-ABI host checks passed" in r.stdout
    assert r.stdout.count("bit-exact") == 3
