#!/usr/bin/env python
"""bench.py with another row threshold for the (384, 768) fused MLP (A/B of engine.Plan.FUSED_MLP_MIN_ROWS on one box):
    python tools/bench_min_rows.py <rows> [bench.py arguments]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
from lvae.engine import Plan  # noqa: E402

Plan.FUSED_MLP_MIN_ROWS = {(384, 768): int(sys.argv[1])}
sys.argv = [os.path.join(REPO, 'bench.py')] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
