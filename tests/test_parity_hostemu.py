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


@pytest.mark.parametrize("name", ["ppp_weno_ab2", "ppp_weno_visc_ab2", "ppb_weno_full", "ppb_amd_config3"])
def test_x_tiled_tendency_kernel(ocn, backend, name, monkeypatch):
    """The kernel for rows wider than a workgroup (Nx > 256), forced onto small grids with two x-tiles per row."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    monkeypatch.setenv("OCNHIP_FUSED_XT", "1")
    worst = run_case(ocn, name)
    bad = {k: v for k, v in worst.items() if v > 2e-11}
    assert not bad, bad


def test_config1_128x128_matches_oracle_hostemu(ocn, backend):
    """BASELINE config 1 at its own 128 x 128 through the host emulation of the general kernels (two steps of the example's
    dt = 0.2); the `-m gpu` twin runs four steps with and without whole-step hipGraphs."""
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    from parity_cases import run_config1
    run_config1(ocn, 0.2, steps=2)


def test_two_slab_models_share_a_context_hostemu(ocn, backend, monkeypatch):
    if backend != "hostemu":
        pytest.skip("host-emulation run only")
    monkeypatch.setenv("OCNHIP_FORCE_DIST", "1")
    monkeypatch.setenv("OCNHIP_OVERLAP", "1")
    from parity_cases import run_two_slab_models_on_one_context
    run_two_slab_models_on_one_context(ocn)
