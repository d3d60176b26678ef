import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_index():
    return load_golden("index_ops.npz")


@pytest.fixture(scope="session")
def golden_blocks():
    return load_golden("blocks.npz")


@pytest.fixture(scope="session")
def golden_fuse():
    return load_golden("fuse.npz")


@pytest.fixture(scope="session")
def golden_cls():
    return load_golden("cls_model.npz")


@pytest.fixture(scope="session")
def golden_seg():
    return load_golden("seg_model.npz")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


@pytest.fixture(scope="session")
def golden_round2():
    return load_golden("round2.npz")
