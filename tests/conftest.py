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


def load_seeded_into(model, sd_np):
    """Load seeded numpy weights (reference key names) into a product model, keeping its own buffers."""
    import torch
    full = model.state_dict()
    for k, v in sd_np.items():
        assert k in full and tuple(full[k].shape) == tuple(v.shape), k
        full[k] = torch.from_numpy(v)
    model.load_state_dict(full)
    return model


@pytest.fixture(scope='session')
def product_model(qarv_seeded_sd):
    """qarv_base on cuda:0 with the seeded weights, in compress mode (HIP path)."""
    import torch
    import lvae
    assert torch.cuda.is_available()
    m = lvae.get_model('qarv_base')
    load_seeded_into(m, qarv_seeded_sd)
    m = m.to('cuda:0')
    m.eval()
    m.compress_mode()
    return m
