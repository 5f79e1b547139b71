"""CPU-side parity: the *same kernel sources* as libocnhip.so, compiled for the host (csrc/compat.h,
OCN_HOST_EMU), against the oracle.  Catches indexing / arithmetic errors without a GPU; the GPU run
of the same cases is tests/test_parity_gpu.py."""
import pytest

from parity_cases import CASES, run_case


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_oracle(ocn, backend, name):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    worst = run_case(ocn, name)
    bad = {k: v for k, v in worst.items() if v > CASES[name].get("tol", 2e-11)}
    assert not bad, bad
