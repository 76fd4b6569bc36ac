import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
# The CPU side of the tests (oracle runs, fp64 checks) is many small matrix products: on a 128-core GPU box the default
# thread pools spend their time spinning (and oversubscribe under pytest-xdist).  Eight threads per process is faster.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")


def pytest_configure(config):
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference mount (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/wild_completion")
    skip_ref = pytest.mark.skip(reason="reference mount absent (GPU box)")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
