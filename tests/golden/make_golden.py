"""Regenerates tests/golden/*.npz: final interior fields of a few parity cases computed by the NumPy oracle.

These are NOT outputs of the reference (no Julia toolchain here, and the reference's regression data are remote
DataDeps): they freeze the oracle's own answers so that (a) an accidental change of the oracle shows up as a diff
against committed data and (b) the HIP path can be checked on the GPU box against numbers that were produced in a
different process, on a different machine, by different code.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as O                      # noqa: E402
from parity_cases import CASES, build   # noqa: E402

GOLDEN_CASES = ["ppp_weno_ab2", "ppp_weno_rk3_2tracers", "ppb_amd_config3", "bbb_weno_walls", "bfb_weno_slice"]


def final_fields(mod, name):
    cfg = CASES[name]
    m = build(mod, cfg)
    for _ in range(cfg["steps"]):
        mod.time_step(m, cfg["dt"])
    out = {"u": m.u.interior(), "v": m.v.interior(), "w": m.w.interior(), "pNHS": m.pNHS.interior()}
    for n, f in m.tracers.items():
        out["tracer_" + n] = f.interior()
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


if __name__ == "__main__":
    for name in GOLDEN_CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **final_fields(O, name))
        print("wrote", name)
