"""not-gpu: the C-ABI shared library loads here (no GPU) and exports exactly the symbols include/lvae_hip.h declares."""
import os
import re
import subprocess

import pytest

from lvae import _native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, 'include', 'lvae_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return set(re.findall(r'\b(lvae_[a-z0-9_]+)\s*\(', src))


def test_library_loads_and_reports_abi():
    L = _native.lib()
    assert L.lvae_abi_version() == _native.ABI_VERSION
    assert b'gfx950' in L.lvae_build_info()


def test_exports_match_header():
    declared = _declared()
    out = subprocess.check_output(['nm', '-D', '--defined-only', _native.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln and ln.split()[-1].startswith('lvae_')}
    assert declared == exported, (declared - exported, exported - declared)
    assert declared == set(_native.SIGNATURES), (declared ^ set(_native.SIGNATURES))


def test_gemm_desc_layout_matches_header():
    """ctypes mirror of lvae_gemm_desc: same field order as the header (a mismatch would silently corrupt launches)."""
    src = open(os.path.join(REPO, 'include', 'lvae_hip.h')).read()
    body = src[src.index('typedef struct {'):src.index('} lvae_gemm_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for stmt in body.split('{', 1)[1].split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(','):
            names.append(re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*$', part.strip())[0])
    assert names == [f[0] for f in _native.GemmDesc._fields_]


def test_argument_validation_without_gpu():
    """Host-side argument checks return -22 before any HIP call (safe on a GPU-less box)."""
    L = _native.lib()
    assert L.lvae_gemm_f32(None, None) == -22
    assert L.lvae_dwconv_ln_f32(None, None, None, None, None, None, None, None, 1, 1, 1, 128, 7, None) == -22
    assert L.lvae_gemv_f32(None, None, None, None, 4, 4, 0, 0, None) == -22
    assert L.lvae_stem_f32(None, None, None, None, 1, 64, 64, 192, 0.0, 1.0, None, None) == -22
    assert L.lvae_range_flag_f32(None, 16, 0.0, 1.0, None, None) == -22


def test_c99_client_round_trips_the_coder(tmp_path):
    """The boundary from C: tests/c_client/coder_roundtrip.c includes include/lvae_hip.h as pedantic C99 (no C++ in the header), links
    liblvae_hip.so and round-trips 100 000 symbols (escapes included) through lvae_build_gaussian_tables / lvae_rans_encode_with_indexes /
    lvae_rans_decode_with_indexes, then checks the documented error codes (-2 short buffer, -4 bad index, < 0 truncated stream)."""
    import shutil
    import subprocess
    from lvae import _native
    if shutil.which('gcc') is None:
        pytest.skip('no gcc on this host')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(os.path.abspath(_native.LIB_PATH))
    exe = str(tmp_path / 'coder_rt')
    cc = subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-O1', '-I' + os.path.join(root, 'include'),
                         os.path.join(root, 'tests', 'c_client', 'coder_roundtrip.c'), '-o', exe, '-L' + libdir, '-l:' + os.path.basename(_native.LIB_PATH),
                         '-lm', '-Wl,-rpath,' + libdir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    tag, abi, nbytes = run.stdout.split()
    assert tag == 'ok' and int(abi) == _native.ABI_VERSION and int(nbytes) > 8
