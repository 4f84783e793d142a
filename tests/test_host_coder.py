"""not-gpu: the product's C++ host coder (liblvae_hip.so, C ABI) against the oracle restatement, bit for bit."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import compressai_semantics as cs
from oracle.qarv_oracle import DiscretizedGaussianOracle


@pytest.fixture(scope='module')
def L():
    from lvae import _native
    return _native.lib()


@pytest.fixture(scope='module')
def tabs():
    d = DiscretizedGaussianOracle()
    d.update()
    return (np.ascontiguousarray(d._quantized_cdf.numpy().astype(np.int32)),
            np.ascontiguousarray(d._cdf_length.numpy().astype(np.int32)),
            np.ascontiguousarray(d._offset.numpy().astype(np.int32)), d)


def _enc(L, sym, idx, tabs):
    cdf, ln, off, _ = tabs
    out = np.empty(8 * sym.size + 64, dtype=np.uint8)
    n = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, sym.size, cdf.ctypes.data, cdf.shape[1],
                                        ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
    assert n >= 8
    return out[:n].tobytes()


def _dec(L, s, idx, tabs):
    cdf, ln, off, _ = tabs
    buf = np.frombuffer(s, dtype=np.uint8)
    out = np.empty(idx.size, dtype=np.int32)
    rc = L.lvae_rans_decode_with_indexes(buf.ctypes.data, buf.size, idx.ctypes.data, idx.size, cdf.ctypes.data, cdf.shape[1],
                                         ln.ctypes.data, off.ctypes.data, out.ctypes.data)
    return rc, out


@pytest.mark.parametrize('n,spread', [(0, 1.0), (1, 1.0), (7, 1.0), (4097, 0.3), (100000, 1.5), (30000, 40.0)])
def test_streams_match_oracle(L, tabs, n, spread):
    cdf, ln, off, dg = tabs
    g = np.random.default_rng(n + 1)
    idx = g.integers(0, 64, size=n).astype(np.uint8)
    sym = np.rint(g.normal(0, 1, size=n) * dg.scale_table.numpy()[idx] * spread).astype(np.int32)   # spread 40 => many escapes
    s = _enc(L, sym, idx, tabs)
    ref = cs.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist())
    assert s == ref
    rc, out = _dec(L, s, idx, tabs)
    assert rc == 0 and np.array_equal(out, sym)
    assert cs.RansDecoder().decode_with_indexes(s, idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist()) == sym.tolist()


def test_escape_extremes(L, tabs):
    sym = np.array([0, 2 ** 27 - 300, -(2 ** 27 - 300), 70000, -70000, 3, -3, 5000, -5000], dtype=np.int32)
    idx = np.array([0, 63, 63, 0, 0, 31, 31, 7, 7], dtype=np.uint8)
    cdf, ln, off, _ = tabs
    s = _enc(L, sym, idx, tabs)
    assert s == cs.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist())
    rc, out = _dec(L, s, idx, tabs)
    assert rc == 0 and np.array_equal(out, sym)


def test_decode_rejects_garbage(L, tabs):
    idx = np.zeros(1000, dtype=np.uint8)
    rc, _ = _dec(L, b'\x00' * 6, idx, tabs)            # too short / not a multiple of 4
    assert rc < 0
    rc, _ = _dec(L, b'\xff' * 8, idx, tabs)            # truncated: runs off the stream
    assert rc < 0


def test_small_output_buffer(L, tabs):
    cdf, ln, off, _ = tabs
    sym, idx = np.zeros(10000, dtype=np.int32), np.full(10000, 63, dtype=np.uint8)
    out = np.empty(16, dtype=np.uint8)
    n = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, sym.size, cdf.ctypes.data, cdf.shape[1],
                                        ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
    assert n == -2


def test_batch_api_equals_single(L, tabs):
    from lvae.models.entropy_coding import rans_decode_streams, rans_encode_streams
    cdf, ln, off, dg = tabs
    g = np.random.default_rng(9)
    syms, idxs = [], []
    for n in (3072, 12288, 0, 147456, 49152, 5):
        idx = g.integers(0, 64, size=n).astype(np.uint8)
        syms.append(np.rint(g.normal(0, 1, size=n) * dg.scale_table.numpy()[idx]).astype(np.int32)); idxs.append(idx)
    for nt in (1, 3, 0):
        strings = rans_encode_streams((cdf, ln, off), syms, idxs, nt)
        assert strings == [_enc(L, s, i, tabs) for s, i in zip(syms, idxs)]
        outs = [np.empty(s.size, dtype=np.int32) for s in syms]
        rans_decode_streams((cdf, ln, off), strings, idxs, outs, nt)
        assert all(np.array_equal(a, b) for a, b in zip(outs, syms))
    with pytest.raises(ValueError):
        rans_decode_streams((cdf, ln, off), [b'\xff' * 8], [np.zeros(100, np.uint8)], [np.empty(100, np.int32)], 1)


def test_shared_decode_tables_under_contention(L, tabs):
    """The decoder's per-row bucket tables are one set per batch call, built lazily by whichever stream touches a row first and read
    by all of them (csrc/rans_host.cpp: LvaeDecTabs).  Many short streams that all start on the same rows, decoded by many threads,
    repeatedly: every stream must decode exactly as it does alone; an invalid table row must fail every stream that touches it."""
    from lvae.models.entropy_coding import rans_decode_streams, rans_encode_streams
    cdf, ln, off, dg = tabs
    g = np.random.default_rng(21)
    syms, idxs = [], []
    for k in range(96):
        n = int(g.integers(1, 400))
        idx = np.concatenate([np.arange(64, dtype=np.uint8)[:min(n, 64)], g.integers(0, 64, size=max(0, n - 64)).astype(np.uint8)])
        syms.append(np.rint(g.normal(0, 1, size=n) * dg.scale_table.numpy()[idx]).astype(np.int32)); idxs.append(idx)
    strings = rans_encode_streams((cdf, ln, off), syms, idxs, 0)
    alone = [_dec(L, s, i, tabs) for s, i in zip(strings, idxs)]
    assert all(rc == 0 and np.array_equal(o, s) for (rc, o), s in zip(alone, syms))
    for rep in range(40):
        outs = [np.full(s.size, -12345, dtype=np.int32) for s in syms]
        rans_decode_streams((cdf, ln, off), strings, idxs, outs, 0 if rep % 2 else 16)
        assert all(np.array_equal(a, b) for a, b in zip(outs, syms)), rep
    bad_ln = ln.copy()
    bad_ln[5] = 1                                                    # a cdf row of length 1: not a distribution
    with pytest.raises(ValueError):
        rans_decode_streams((cdf, bad_ln, off), strings, idxs, [np.empty(s.size, dtype=np.int32) for s in syms], 16)


def test_pmf_to_quantized_cdf_matches_oracle(L):
    g = np.random.default_rng(1)
    for n in (2, 5, 64, 248):
        for _ in range(25):
            p = (g.random(n) ** 6).astype(np.float32)
            p /= p.sum()
            out = np.zeros(n + 1, dtype=np.uint32)
            assert L.lvae_pmf_to_quantized_cdf(p.ctypes.data, n, 16, out.ctypes.data) == 0
            assert out.tolist() == cs.pmf_to_quantized_cdf(p.tolist(), 16)
    bad = np.array([0.5, -0.1], dtype=np.float32)
    out = np.zeros(3, dtype=np.uint32)
    assert L.lvae_pmf_to_quantized_cdf(bad.ctypes.data, 2, 16, out.ctypes.data) == -1
    z = np.zeros(4, dtype=np.float32)
    out = np.zeros(5, dtype=np.uint32)
    assert L.lvae_pmf_to_quantized_cdf(z.ctypes.data, 4, 16, out.ctypes.data) == -2


def test_product_update_matches_reference_tables(golden_dir):
    """lvae.models.entropy_coding.DiscretizedGaussian.update() (torch pmf + native quantiser) == the tables the
    REFERENCE's DiscretizedGaussian produced (tests/golden/discretized_gaussian_tables.npz), bit for bit."""
    from lvae.models.entropy_coding import DiscretizedGaussian
    g = np.load(os.path.join(golden_dir, 'discretized_gaussian_tables.npz'))
    dg = DiscretizedGaussian(cdf_form='erf')
    dg.update()
    assert np.array_equal(dg.scale_table.numpy(), g['scale_table'])
    assert np.array_equal(dg._quantized_cdf.numpy(), g['quantized_cdf'])
    assert np.array_equal(dg._cdf_length.numpy(), g['cdf_length'])
    assert np.array_equal(dg._offset.numpy(), g['offset'])
    q, l, o = dg.host_tables()
    assert q.dtype == np.int32 and q.flags['C_CONTIGUOUS'] and q.shape == (64, 249)


@pytest.mark.parametrize('form,name', [(0, 'discretized_gaussian_tables'), (1, 'gaussian_conditional_tables')])
def test_native_table_builder_equals_reference_tables(L, golden_dir, form, name):
    """lvae_build_gaussian_tables (host C++, erf/erfc correctly rounded from double -- what DiscretizedGaussian.update() calls, on
    whatever device the module lives) == the tables the REFERENCE's classes built (DiscretizedGaussian: erf form, QARV; stock
    CompressAI-style GaussianConditional: erfc form, QRes), bit for bit; plus the CDF-row invariants of SURVEY.md 8(c)."""
    import scipy.stats
    g = np.load(os.path.join(golden_dir, f'{name}.npz'))
    table = np.ascontiguousarray(g['scale_table'].astype(np.float32))
    q = np.zeros((64, 256), dtype=np.int32); ln = np.zeros(64, dtype=np.int32); off = np.zeros(64, dtype=np.int32)
    mx = L.lvae_build_gaussian_tables(table.ctypes.data, 64, float(-scipy.stats.norm.ppf(0.5e-9)), form, q.ctypes.data, 256,
                                      ln.ctypes.data, off.ctypes.data)
    assert mx == 249
    assert np.array_equal(ln, g['cdf_length']) and np.array_equal(off, g['offset'])
    assert np.array_equal(q[:, :249], g['quantized_cdf']) and not q[:, 249:].any()
    for i in range(64):
        row = q[i, :ln[i]]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) >= 1)
    # too small a row stride is refused, not overrun
    assert L.lvae_build_gaussian_tables(table.ctypes.data, 64, float(-scipy.stats.norm.ppf(0.5e-9)), form, q.ctypes.data, 100,
                                        ln.ctypes.data, off.ctypes.data) == -2


def test_update_is_device_independent(golden_dir, monkeypatch):
    """update() never evaluates the pmf with torch ops on the module's device (a GPU's erf may differ in the last ulp and a
    stream must decode anywhere): the erfc-form module equals the reference table too, and `.to()` keeps the tables."""
    from lvae.models.entropy_coding import DiscretizedGaussian
    g = np.load(os.path.join(golden_dir, 'gaussian_conditional_tables.npz'))
    dg = DiscretizedGaussian(cdf_form='erfc', scale_bound=0.11)
    assert dg.update_scale_table(torch.from_numpy(g['scale_table']))
    assert np.array_equal(dg._quantized_cdf.numpy(), g['quantized_cdf']) and np.array_equal(dg._offset.numpy(), g['offset'])
    assert np.array_equal(dg._cdf_length.numpy(), g['cdf_length'])
    q, l, o = dg.host_tables()
    assert q.dtype == np.int32 and q.flags['C_CONTIGUOUS'] and np.array_equal(q, g['quantized_cdf'])
    assert not dg.update_scale_table(torch.from_numpy(g['scale_table']) * 2)       # tables exist: no-op unless forced
    def boom(*a, **k):
        raise AssertionError('update() must not evaluate the pmf with torch ops')
    monkeypatch.setattr(torch, 'erf', boom); monkeypatch.setattr(torch, 'erfc', boom)
    dg2 = DiscretizedGaussian(cdf_form='erf')
    dg2.update()
    assert dg2._quantized_cdf.shape == (64, 249)


def test_scale_tables_are_host_independent_and_equal_the_reference(golden_dir):
    """log_spaced_table pins every rounding of the reference's torch.exp(torch.linspace(...)) scale tables (vectorised torch kernels
    differ in the last ulp between CPUs -- seen between the build container and the MI355X box's host): all three tables the codecs
    use equal the reference run's bit for bit, product and oracle alike."""
    from lvae.models.entropy_coding import log_spaced_table
    for fn in (log_spaced_table, cs.log_spaced_table):
        assert np.array_equal(fn(0.11, 20.0, 64).numpy(), np.load(os.path.join(golden_dir, 'discretized_gaussian_tables.npz'))['scale_table'])
        assert np.array_equal(fn(0.1, 20, 64).numpy(), np.load(os.path.join(golden_dir, 'gaussian_conditional_tables.npz'))['scale_table'])
        assert np.array_equal(fn(0.11, 20, 128).numpy(), np.load(os.path.join(golden_dir, 'qres34m_lossless_64x128.npz'))['scale_table'])


def test_reciprocal_encoder_equals_division_form(L):
    """The encoder's division-free step (Alverson reciprocal, csrc/rans_host.cpp::enc_put_ent) against the published formula
    x' = ((x / freq) << 16) + (x % freq) + start, state by state: every frequency 1 .. 65535 at the edges of the state range
    [2^31, 2^47 * freq) before renormalisation and [2^31 .. 2^63) overall, plus random states.  (The stream tests above compare whole
    streams with the oracle coder, which divides.)"""
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    g = np.random.default_rng(7)

    def check(x, start, freq):
        rc = L.lvae_rans_enc_step_selftest(ctypes.c_uint64(x), start, freq, ctypes.byref(a), ctypes.byref(b))
        assert rc == 0 and a.value == b.value, (x, start, freq, a.value, b.value)
        xr = x >> 32 if x >= (freq << 47) else x                                  # the published renormalisation + step
        assert a.value == ((xr // freq) << 16) + (xr % freq) + start

    freqs = list(range(1, 300)) + [2 ** k + d for k in range(8, 16) for d in (-1, 0, 1)] + [65535, 65534, 43691, 21845, 33333] + \
        [int(v) for v in g.integers(300, 65535, size=400)]
    for freq in freqs:
        if not 1 <= freq <= 65535:
            continue
        hi = freq << 47                                                           # x_max: states at / above it are renormalised first
        xs = [1 << 31, (1 << 31) + 1, hi - 1, hi, hi + 1, (1 << 63) - 1, (1 << 63) - freq, (1 << 62) + 12345, hi - freq, hi - freq - 1,
              (hi // freq) * freq, (hi // freq) * freq - 1]
        xs += [int(v) for v in g.integers(1 << 31, 1 << 63, size=40, dtype=np.uint64)]
        xs += [int(v) % hi for v in g.integers(1 << 31, 1 << 63, size=20, dtype=np.uint64) if int(v) % hi >= (1 << 31)]
        for x in xs:
            if (1 << 31) <= x < (1 << 63):
                check(x, int(g.integers(0, 65536 - freq + 1)), freq)


def _oracle_dec(s, idx, cdf, ln, off):
    return np.asarray(cs.RansDecoder().decode_with_indexes(s, idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist()), dtype=np.int32)


def test_most_probable_symbol_path_on_off_and_switching(L, tabs):
    """csrc/rans_host.cpp::decode_stream takes the row's most probable symbol without a table load when the state falls inside it,
    and switches that path off (and probes again, with back-off) on streams that are not mostly modes.  One stream that goes
    compressible -> noise -> compressible -> escapes -> compressible over many 1024-symbol windows crosses every state of that
    switch; the decoded symbols must be the oracle decoder's whatever the path did."""
    cdf, ln, off, dg = tabs
    g = np.random.default_rng(77)
    sc = dg.scale_table.numpy()
    parts = []
    for kind, n in (('zeros', 5000), ('noise', 23000), ('zeros', 9000), ('escapes', 7000), ('peaked', 30000), ('noise', 3000), ('zeros', 2049)):
        idx = g.integers(0, 64, size=n).astype(np.uint8)
        if kind == 'zeros':
            sym = np.zeros(n, np.int32)
        elif kind == 'peaked':                         # ~95 % modes
            sym = np.where(g.random(n) < 0.95, 0, np.rint(g.normal(0, 1, n) * sc[idx])).astype(np.int32)
        elif kind == 'noise':
            sym = np.rint(g.normal(0, 1, n) * sc[idx] * 1.5).astype(np.int32)
        else:
            sym = np.rint(g.normal(0, 1, n) * sc[idx] * 60.0).astype(np.int32)
        parts.append((idx, sym))
    idx = np.concatenate([p[0] for p in parts]); sym = np.concatenate([p[1] for p in parts])
    s = _enc(L, sym, idx, tabs)
    rc, out = _dec(L, s, idx, tabs)
    assert rc == 0 and np.array_equal(out, sym)
    assert np.array_equal(_oracle_dec(s, idx, cdf, ln, off), sym)
    # a truncated stream still fails cleanly on the fast path (all modes: every symbol takes it)
    idx0 = np.full(40000, 30, np.uint8); sym0 = np.zeros(40000, np.int32)
    s0 = _enc(L, sym0, idx0, tabs)
    rc, _ = _dec(L, s0[:len(s0) // 2 // 4 * 4], idx0, tabs)
    assert rc != 0


def test_most_probable_symbol_path_on_odd_tables(L):
    """Rows the Gaussian tables never produce: the mode at the table's first / last own symbol, a row of one own symbol, an escape of frequency 1, a mode of frequency 65535, equal frequencies (first one wins: any choice decodes the same)."""
    rows = [
        [0, 60000, 62000, 65000, 65536],               # mode first, 3 own symbols + escape
        [0, 100, 1000, 65000, 65536],                  # mode last own symbol
        [0, 65535, 65536],                             # one own symbol of frequency 65535 + escape
        [0, 32768, 65535, 65536],                      # two own symbols, an escape of frequency 1
        [0, 16384, 32768, 49152, 65536],               # flat
        [0, 1, 2, 3, 65533, 65534, 65535, 65536],      # narrow tails around a huge mode
    ]
    stride = 16
    cdf = np.zeros((len(rows), stride), np.int32)
    ln = np.zeros(len(rows), np.int32)
    for r, row in enumerate(rows):
        cdf[r, :len(row)] = row; ln[r] = len(row)
    off = np.array([0, -1, 0, 0, -2, -3], np.int32)
    g = np.random.default_rng(5)
    n = 20000
    idx = g.integers(0, len(rows), size=n).astype(np.uint8)
    own = ln[idx] - 2                                   # own symbols per row (the escape excluded)
    inside = g.random(n) < 0.9
    k = (g.random(n) * np.maximum(own, 1)).astype(np.int32)
    sym = np.where(inside & (own > 0), k + off[idx], g.integers(-40, 40, size=n)).astype(np.int32)
    out = np.empty(8 * n + 64, dtype=np.uint8)
    nb = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, n, cdf.ctypes.data, stride, ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
    assert nb >= 8
    s = out[:nb].tobytes()
    assert s == cs.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist())
    dec = np.empty(n, np.int32)
    buf = np.frombuffer(s, dtype=np.uint8)
    rc = L.lvae_rans_decode_with_indexes(buf.ctypes.data, buf.size, idx.ctypes.data, n, cdf.ctypes.data, stride, ln.ctypes.data, off.ctypes.data, dec.ctypes.data)
    assert rc == 0 and np.array_equal(dec, sym)
    assert np.array_equal(_oracle_dec(s, idx, cdf, ln, off), sym)


@pytest.mark.parametrize('cdf_len,ok', [(257, True), (258, False), (3, True), (2, False)])
def test_row_length_limit_is_the_same_for_encoder_and_decoder(L, cdf_len, ok):
    """one limit for the legal length of a CDF row on both sides (3 <= cdf_len <= 257 for the encoder; the decoder's symbol
    ids are bytes).  A 257-entry row (255 own symbols + escape) round-trips and equals the oracle's bytes; a 258-entry row, which the
    encoder used to take and the decoder to refuse, is refused by both (-4); a 2-entry row (escape only) is not writable here."""
    n_int = cdf_len - 1                                   # intervals: own symbols + the escape symbol
    cdf = np.zeros((1, 264), np.int32)
    edges = np.linspace(0, 65536, n_int + 1).astype(np.int64)
    edges[-1] = 65536
    cdf[0, :cdf_len] = edges
    ln, off = np.array([cdf_len], np.int32), np.array([-5], np.int32)
    g = np.random.default_rng(cdf_len)
    n = 5000
    idx = np.zeros(n, np.uint8)
    sym = g.integers(-12, n_int + 4, size=n).astype(np.int32)      # own symbols and escapes on both sides
    out = np.empty(8 * n + 64, dtype=np.uint8)
    nb = L.lvae_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, n, cdf.ctypes.data, cdf.shape[1], ln.ctypes.data, off.ctypes.data, out.ctypes.data, out.size)
    if not ok:
        assert nb == -4
        junk = np.zeros(64, np.uint8)
        dec = np.empty(n, np.int32)
        rc = L.lvae_rans_decode_with_indexes(junk.ctypes.data, junk.size, idx.ctypes.data, n, cdf.ctypes.data, cdf.shape[1], ln.ctypes.data, off.ctypes.data, dec.ctypes.data)
        assert rc == (-4 if cdf_len > 257 else rc)              # (the decoder also reads the one-interval row the published coder can write)
        return
    assert nb >= 8
    s = out[:nb].tobytes()
    assert s == cs.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), ln.tolist(), off.tolist())
    dec = np.empty(n, np.int32)
    buf = np.frombuffer(s, dtype=np.uint8)
    rc = L.lvae_rans_decode_with_indexes(buf.ctypes.data, buf.size, idx.ctypes.data, n, cdf.ctypes.data, cdf.shape[1], ln.ctypes.data, off.ctypes.data, dec.ctypes.data)
    assert rc == 0 and np.array_equal(dec, sym)
