"""pytest configuration: registers the `gpu` marker and puts the repo + product package on sys.path.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks, gloo tests.
`-m gpu` runs on an MI355X box: HIP-path parity tests (call through the C-ABI library).
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'lossy-vae_amd')
for p in (REPO, PKG, os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')

if os.environ.get('LVAE_LIB'):       # study builds only (tools/build_exp.sh, tools/build_gelu_variants.sh): run the suite against another library
    from lvae import _native as _nat
    _nat.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
    print(f'[conftest] LVAE_LIB: testing {_nat.LIB_PATH}', file=sys.stderr)


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


# ----------------------------------------------------------------------------------------------- parity report
# Every golden-parity test records one row per (model, size, lambda, precision) case: symbol flips, scale-index flips, number of
# latent elements, max|dx_hat| against the reference's reconstruction and whether every rANS stream was byte-identical.  The rows are
# printed in pytest's terminal summary (so the driver's log tail shows them even under -q) and written to
# gpurun_out/parity_report.json.
PARITY_ROWS = []


def parity_record(case, sym_flips, idx_flips, n, max_dx, streams_identical, guard=None):
    """guard: the dict tests/parity_util.check_blocks returns (teacher-forced comparison: every flip inside its guard band, every
    element within rounding noise) -- printed with the row."""
    PARITY_ROWS.append(dict(case=case, sym_flips=int(sym_flips), idx_flips=int(idx_flips), n=int(n),
                            max_dx=None if max_dx is None else float(max_dx), streams_identical=bool(streams_identical),
                            guard=guard))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY_ROWS:
        return
    import json
    tr = terminalreporter
    tr.section('parity vs reference goldens: case sym_flips/idx_flips/n max|dx| streams')
    for r in PARITY_ROWS:
        dx = 'n/a' if r['max_dx'] is None else f"{r['max_dx']:.2e}"
        tr.write_line(f"{r['case']}: {r['sym_flips']}/{r['idx_flips']}/{r['n']} {dx} {'same' if r['streams_identical'] else 'DIFF'}")
        if r.get('guard'):
            import parity_util
            tr.write_line('    ' + parity_util.describe(r['guard']))
    clean = sum(1 for r in PARITY_ROWS if r['sym_flips'] == 0 and r['idx_flips'] == 0)
    tr.write_line(f'{clean} of {len(PARITY_ROWS)} cases flip-free; worst max|dx| '
                  f"{max((r['max_dx'] or 0.0) for r in PARITY_ROWS):.2e} (bar 1e-4)")
    gs = [r['guard'] for r in PARITY_ROWS if r.get('guard')]
    if gs:
        tr.write_line(f"guard bands: {sum(g['sym_flips'] for g in gs)} symbol + {sum(g['idx_flips'] for g in gs)} index flips in "
                      f"{sum(g['n'] for g in gs)} teacher-forced elements, all flips inside guard band "
                      f"(worst margins: symbol {max(g['worst_sym_margin'] for g in gs):.1e}, index {max(g['worst_idx_margin'] for g in gs):.1e})")
    try:
        out = os.path.join(REPO, 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_report.json'), 'w') as f:
            json.dump(PARITY_ROWS, f, indent=1)
    except OSError:
        pass
