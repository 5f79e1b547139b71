"""One-off longer sweep of the generators of tests/test_fuzz_small_configurations.py through HIP (not part of the test suite):
    python tools/fuzz_gpu.py <first seed> <count> [small|medium]
prints every configuration whose worst relative error against the oracle exceeds 5e-10, and a count."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
import parity_cases as pc  # noqa: E402
from test_fuzz_small_configurations import random_case, random_medium_case  # noqa: E402

ocn = ge.load_package()
first, count = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "medium"
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng((5000 if kind == "medium" else 1000) + seed)
    cfg = random_medium_case(rng) if kind == "medium" else random_case(rng)
    os.environ.pop("OCNHIP_FORCE_DIST", None)
    os.environ.pop("OCNHIP_OVERLAP", None)
    if kind == "medium" and cfg["topo"] == ("Periodic",) * 3 and rng.random() < 0.4 and cfg["size"][2] >= 8:
        os.environ["OCNHIP_FORCE_DIST"] = "1"
        os.environ["OCNHIP_OVERLAP"] = "1"
    pc.CASES["_fuzz"] = cfg
    try:
        worst = pc.run_case(ocn, "_fuzz")
        scale_bad = {k: v for k, v in worst.items() if not v <= (5e-10 if kind == "medium" else 1e-6)}
        if scale_bad:
            bad += 1
            print("MISMATCH", seed, cfg, scale_bad, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR", seed, cfg, repr(e)[:300], flush=True)
print(f"{kind}: seeds {first}..{first + count - 1}, bad = {bad}")
