"""One-off: BASELINE config 2 at its full size (256^3, WENO5, AB2) through libocnhip.so and through the NumPy oracle
on the same seeded input, one time step, field-by-field comparison.  The oracle needs ~3 minutes per step at this
size, which is why the test suite checks 256^3 through properties only; this script records the direct comparison.

    python tools/full_size_parity.py > profiles/r01_full_size_parity.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402
import oracle as O             # noqa: E402

ocn = ge.load_package()
N = (256, 256, 256)
rng = np.random.default_rng(2024)
init = {n: rng.random(N) - 0.5 for n in "uvw"}
kw = dict(size=N, extent=(1, 1, 1), topology=("Periodic",) * 3)
m = ocn.NonhydrostaticModel(ocn.RectilinearGrid(**kw), advection=ocn.WENO5())
ocn.set_model(m, **init)
t0 = time.time()
om = O.NonhydrostaticModel(O.RectilinearGrid(**kw), advection=O.WENO5())
O.set_model(om, **init)
dt = 0.2 / 256 / np.abs(om.u.data).max()
out = {"size": N, "dt": dt, "steps": 2, "rel_err": {}}
for step in range(2):     # the first step is Euler (G^- = 0), the second a genuine AB2 step
    ocn.time_step(m, dt)
    O.time_step(om, dt)
    for name, a, b in (("u", m.u, om.u), ("v", m.v, om.v), ("w", m.w, om.w), ("pNHS", m.pNHS, om.pNHS),
                       ("Gn_u", m.Gn["u"], om.Gn["u"])):
        err = float(np.abs(a.interior() - b.interior()).max() / np.abs(b.interior()).max())
        out["rel_err"][f"step{step + 1}_{name}"] = err
    print(f"step {step + 1} done after {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
out["oracle_seconds"] = time.time() - t0
out["max_rel_err"] = max(out["rel_err"].values())
print(json.dumps(out))
