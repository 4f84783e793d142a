"""pytest configuration: registers the `gpu` marker and puts the repo + product package on sys.path.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks, gloo tests.
`-m gpu` runs on an MI355X box: HIP-path parity tests (call through the C-ABI library).
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'lossy-vae_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (HIP kernels run); skipped by -m "not gpu"')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def qarv_seeded_sd():
    """Seeded synthetic qarv_base weights (numpy, reference key names) -- regenerated, never shipped."""
    import seeded_init
    from oracle import qarv_oracle
    arch = qarv_oracle.qarv_base_arch()
    return seeded_init.seeded_state_dict(qarv_oracle.qarv_param_shapes(arch), seed=0)
