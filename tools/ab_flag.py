#!/usr/bin/env python3
"""Same-box A/B of engine-level switches under bench.py:  python tools/ab_flag.py NAME=value[,NAME=value...] [bench.py arguments]
(sets contrastive_lift_amd.engine.NAME before bench.main() runs; prints bench.py's JSON line)."""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contrastive_lift_amd import engine
for kv in sys.argv[1].split(","):
    if kv:
        k, v = kv.split("=")
        assert hasattr(engine, k), k
        setattr(engine, k, eval(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
