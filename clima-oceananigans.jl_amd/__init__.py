"""ocnhip: MI355X-native `time_step!` for Oceananigans' NonhydrostaticModel (host-side mirror).

The reference is Julia; no Julia toolchain exists in the build image, so the host side above the
C ABI (include/ocnhip.h) is this thin Python mirror of the reference's constructors and verbs:
``RectilinearGrid``, ``NonhydrostaticModel``, ``set_model`` (= ``set!``), ``time_step`` (= ``time_step!``),
``update_state``, ``fill_halo_regions`` ... with the reference's names, argument meaning and error
behaviour.  The Julia `ROCmGPU` shim a maintainer would add is in INTEGRATION.md / julia/ROCmGPU.jl.
All numerical work happens in libocnhip.so (hand-written HIP for gfx950 + hipFFT/rocFFT + RCCL).
"""
from .api import (Context, RectilinearGrid, NonhydrostaticModel, Periodic, Bounded, Flat, Center, Face,  # noqa: F401
                  WENO5, NoAdvection, CenteredSecondOrder, CenteredFourthOrder, UpwindBiasedFifthOrder, UpwindBiasedFirstOrder, UpwindBiasedThirdOrder, ScalarDiffusivity,
                  AnisotropicMinimumDissipation, FPlane, BuoyancyTracer, SeawaterBuoyancy,
                  FluxBC, ValueBC, GradientBC, Field, CenterField, time_step, set_model, update_state, OcnError)
from . import _lib  # noqa: F401
from . import hydrostatic  # noqa: F401   SplitExplicitFreeSurface, LatitudeLongitudeGrid (BASELINE config 5, first slice)

__all__ = ["Context", "RectilinearGrid", "NonhydrostaticModel", "time_step", "set_model"]
