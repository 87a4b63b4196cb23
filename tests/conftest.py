import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    # The CPU oracle is most of the parity tests' time.  torch sizes its thread pool by the CPUs it can SEE; a GPU box shows all of the
    # host's hardware threads but grants a 1-GPU lease a share of 16 cores, and an oversubscribed pool runs the oracle several times
    # slower (bench.py's cpu_baseline caps its threads the same way).
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 16)))


class Golden:
    """Prefix view over an .npz of reference-generated fixtures (oracle/make_golden.py)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def t(self, key, dtype=None):
        a = self.z[key]
        t = torch.from_numpy(np.array(a))
        return t if dtype is None else t.to(dtype)

    def sub(self, prefix):
        """{name: tensor} of all arrays below `prefix/`."""
        pre = prefix.rstrip("/") + "/"
        return {k[len(pre):]: torch.from_numpy(np.array(self.z[k])) for k in self.z.files
                if k.startswith(pre) and self.z[k].dtype.kind in "fiub"}

    def json(self, key):
        return json.loads(str(self.z[key]))


@pytest.fixture(scope="session")
def g_ops():
    return Golden("ops.npz")


@pytest.fixture(scope="session")
def g_e2e():
    return Golden("e2e_tiny.npz")


@pytest.fixture(scope="session")
def g_masks():
    return Golden("masks.npz")


def gpu_available():
    return torch.cuda.is_available()
