"""Self-consistency pins for the plain-C rANS / CDF restatement (oracle/rans_oracle.c): CompressAI is absent
(parity unpinned), so: (i) round trips incl. bypass escapes, (ii) CDF invariants, (iii) coded size vs entropy."""
import numpy as np
import pytest
import torch

from oracle import compressai_semantics as cs
from oracle.qarv_oracle import DiscretizedGaussianOracle


@pytest.fixture(scope='module')
def dg():
    d = DiscretizedGaussianOracle()
    d.update()
    return d


def _tabs(dg):
    return dg._quantized_cdf.tolist(), dg._cdf_length.tolist(), dg._offset.tolist()


def test_pmf_to_quantized_cdf_invariants():
    g = np.random.default_rng(0)
    for n in (2, 3, 17, 247):
        for _ in range(20):
            p = g.random(n).astype(np.float32) ** 8      # many near-zero entries -> exercises the steal loop
            p /= p.sum()
            cdf = np.array(cs.pmf_to_quantized_cdf(p.tolist(), 16))
            assert cdf[0] == 0 and cdf[-1] == 65536 and len(cdf) == n + 1
            assert np.all(np.diff(cdf) >= 1)
    # known answer: uniform pmf of 4 -> equal quarters
    assert cs.pmf_to_quantized_cdf([0.25] * 4, 16) == [0, 16384, 32768, 49152, 65536]
    # a zero-probability symbol steals one count from the smallest freq>1 symbol to its right
    assert cs.pmf_to_quantized_cdf([0.0, 0.5, 0.5], 16) == [0, 1, 32768, 65536]
    with pytest.raises(ValueError):
        cs.pmf_to_quantized_cdf([-0.1, 1.1], 16)


@pytest.mark.parametrize('n', [1, 2, 63, 1000, 50000])
def test_round_trip_random(dg, n):
    g = np.random.default_rng(n)
    idx = g.integers(0, 64, size=n)
    sym = np.rint(g.normal(0, 1, size=n) * dg.scale_table.numpy()[idx] * 1.5).astype(np.int32)
    s = cs.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), *_tabs(dg))
    assert len(s) % 4 == 0 and len(s) >= 8
    out = cs.RansDecoder().decode_with_indexes(s, idx.tolist(), *_tabs(dg))
    assert out == sym.tolist()


def test_round_trip_escapes(dg):
    # far-out symbols force the bypass path, n_bypass up to 8 nibbles (|v| < 2^27 is the range the upstream shift logic supports)
    sym = [0, 1, -1, 5000, -5000, 2 ** 20, -(2 ** 20), 123456789, -123456789, 7, -7, 2 ** 27 - 300, -(2 ** 27 - 300)]
    idx = [0, 63, 0, 0, 0, 5, 5, 63, 63, 10, 10, 1, 1]
    s = cs.RansEncoder().encode_with_indexes(sym, idx, *_tabs(dg))
    assert cs.RansDecoder().decode_with_indexes(s, idx, *_tabs(dg)) == sym


def test_empty_stream(dg):
    s = cs.RansEncoder().encode_with_indexes([], [], *_tabs(dg))
    assert len(s) == 8      # just the flushed 64-bit state
    assert cs.RansDecoder().decode_with_indexes(s, [], *_tabs(dg)) == []


def test_coded_size_matches_entropy(dg):
    """(iii) symbols drawn from the model => coded bits within ~1% of sum(-log2 P) of the eval-mode likelihood
    (lvae/models/qarv/model.py:95-96)."""
    g = torch.Generator().manual_seed(0)
    n = 200000
    scales = torch.exp(torch.empty(n).uniform_(np.log(0.11), np.log(12.0), generator=g))
    means = torch.empty(n).uniform_(-3, 3, generator=g)
    x = means + scales * torch.randn(n, generator=g)
    scales, means, x = scales.view(1, -1), means.view(1, -1), x.view(1, -1)
    idx = dg.build_indexes(scales)
    # the coder models each symbol with the TABLE scale (>= true scale): evaluate the likelihood there
    tscale = dg.scale_table[idx.long()]
    dg.eval()
    _, p = dg(x, tscale, means)
    bits = float(-torch.log2(p).sum())
    s = dg.compress(x, idx, means=means)[0]
    assert abs(len(s) * 8 - bits) / bits < 0.01, (len(s) * 8, bits)
    out = dg.decompress([s], idx, means=means)
    assert torch.equal(out, dg.quantize(x, 'dequantize', means))


def test_build_indexes_is_searchsorted(dg):
    """idx = #{i<63: table[i] < max(s, 0.11)} == searchsorted(table[:63], s, 'left')."""
    s = torch.exp(torch.linspace(np.log(0.05), np.log(40.0), 5000))
    s = torch.cat([s, dg.scale_table, dg.scale_table * (1 + 1e-6), dg.scale_table * (1 - 1e-6)])
    ref = dg.build_indexes(s)
    t = dg.scale_table[:-1].contiguous()
    alt = torch.searchsorted(t, torch.max(s, dg.lower_bound_scale.bound), right=False).int()
    assert torch.equal(ref, alt)
    assert int(ref.min()) == 0 and int(ref.max()) == 63
