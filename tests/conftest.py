import os
import sys

import pytest

# torch first: it brings its own copy of the HIP runtime, and when libcalitas_hip.so (linked against /opt/rocm) is loaded before it in
# the same process, torch's device initialisation later answers "no ROCm-capable device".  A full `pytest tests` run imports torch at
# collection anyway (tests/test_distributed_gloo.py); this makes a run of selected files behave the same.  (bench.py imports torch first.)
try:
    import torch  # noqa: F401
except Exception:      # the CPU-only tests of the library do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
