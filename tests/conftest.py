import json
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("DCMT_POISON", "1")   # host entry points poison their output staging buffer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session")
def golden():
    """Small golden vectors (tests/golden/small_cases.npz; made by make_golden.py)."""
    with np.load(os.path.join(GOLDEN_DIR, "small_cases.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN_DIR, "meta.json")) as f:
        return json.load(f)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    neq = bits(a) != bits(b)
    if neq.any():
        idx = np.argwhere(neq)
        first = tuple(idx[0])
        raise AssertionError(f"{what}: {int(neq.sum())} of {a.size} elements differ bitwise; first at {first}: "
                             f"{a[first]!r} vs {b[first]!r}; max |d| = {np.nanmax(np.abs(a - b))}")
