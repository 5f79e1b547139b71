"""python tools/run_smoke.py: __graft_entry__.smoke() (used by tools/gpu_run.sh, whose step arguments cannot hold quotes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.smoke()
