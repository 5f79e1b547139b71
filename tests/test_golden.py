"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
the oracle must still reproduce them, and the C-ABI path (host emulation here, HIP on the GPU box) must match them."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden import GOLDEN_CASES, final_fields   # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(got, name, tol):
    ref = np.load(os.path.join(GOLD, name + ".npz"))
    assert sorted(ref.files) == sorted(got)
    for k in ref.files:
        scale = max(np.abs(ref[k]).max(), 1e-300)
        assert np.abs(got[k] - ref[k]).max() / scale < tol, (name, k)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_reproduces_golden(name):
    import oracle as O
    _check(final_fields(O, name), name, 1e-13)


@pytest.mark.parametrize("name", GOLDEN_CASES[:1] + GOLDEN_CASES[3:4])   # the rest run on the GPU (and through the oracle comparison here)
def test_library_matches_golden_hostemu(ocn, backend, name):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    _check(final_fields(ocn, name), name, 2e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_library_matches_golden_gpu(ocn, backend, name):
    if backend != "gpu":
        pytest.skip("HIP run only")
    _check(final_fields(ocn, name), name, 2e-11)
