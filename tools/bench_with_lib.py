#!/usr/bin/env python
"""bench.py against another build of the native library (A/B of kernel variants on one box):
    python tools/bench_with_lib.py _bin/<name>/liblvae_hip.so [bench.py arguments]
(tools/build_exp.sh makes such builds; the product bench itself has no library switch)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
from lvae import _native  # noqa: E402

_native.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(REPO, 'bench.py')] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
