#!/usr/bin/env python3
"""BASELINE.json configs[4]: two-hand (left + right MANO) batched LBS, 1024 crops per node, all-gather of the
778x3 vertices when N > 1.  Thin alias of `python bench.py --workload mano_lbs` (same launcher: `--gpus N`
starts its own N ranks; under torchrun every process is a rank)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "bench.py"), "--workload", "mano_lbs"] + sys.argv[1:]
if "--steps" not in sys.argv:
    sys.argv += ["--steps", "200", "--warmup", "300"]
runpy.run_path(sys.argv[0], run_name="__main__")
