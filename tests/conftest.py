import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# `BTS_CONV_PRECISION=1 python -m pytest tests -m gpu` runs the whole suite with every convolution in the
# fp32-emulated-on-bf16 arithmetic (bts_amd/ops.py: _ENV_PRECISION; libbts_hip.so reads the same variable).  Tests that pin
# WHICH fp32 kernel family runs (bit-identity of two fp32 kernels, kernel names in the launch trace), and the two
# whole-model training-step tests (the emulated mode is an inference mode: through batch-statistic BN + ReLU the
# 160-layer gradient lands at 8.4e-3 global relative L2 against fp64, the CPU fp32 step at 3.3e-3, the bar at 2x that)
# are skipped in that run.
EMULATED_RUN = os.environ.get("BTS_CONV_PRECISION", "0").strip() == "1"
fp32_only = pytest.mark.skipif(EMULATED_RUN, reason="pins the fp32-MFMA kernel families / training bars; not part of the BTS_CONV_PRECISION=1 run")
