"""CPU oracle for the NonhydrostaticModel ``time_step!`` hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy float64 restatement of the
algorithm of the reference (Oceananigans.jl v0.76.8, pure Julia).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product path (``clima-oceananigans.jl_amd/`` + ``libocnhip.so``)
never does and fails loudly when the HIP library is missing.

Parity status
-------------
The reference cannot be executed here (no ``julia`` binary; its regression data are
remote DataDeps that are not vendored -- ``test/data_dependencies.jl:19-38``), so
this oracle is pinned by the reference's own *known-answer / property tests*
re-implemented in ``tests/test_oracle_*.py`` (SURVEY.md section 8c list 1-13):
Poisson ``lap(phi) == R`` on all supported topologies, second-order convergence to
analytic cosine modes, Thomas-vs-dense, DCT permutation examples, halo-fill
identities, AB2-first-step == Euler, incompressibility, tracer conservation,
Taylor-Green decay, Gaussian advection and WENO5 fifth-order convergence.
WENO5 *numerical values* are not pinned by any stored vector in the reference
(it only smoke-tests WENO5): for those, parity is "pinned by property tests only".

Each function cites the reference file:line it restates (paths relative to
``/root/reference/src``).
"""

from .grid import RectilinearGrid, Periodic, Bounded, Flat, Center, Face  # noqa: F401
from .fields import Field, fill_halo_regions  # noqa: F401
from .model import (NonhydrostaticModel, WENO5, CenteredSecondOrder, CenteredFourthOrder,  # noqa: F401
                    UpwindBiasedFifthOrder, UpwindBiasedFirstOrder, UpwindBiasedThirdOrder, ScalarDiffusivity, FPlane, BuoyancyTracer,
                    SeawaterBuoyancy, AnisotropicMinimumDissipation,
                    FluxBC, ValueBC, GradientBC, time_step, set_model)
from . import poisson  # noqa: F401
from . import split_explicit  # noqa: F401
