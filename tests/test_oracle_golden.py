"""Pins the CPU oracle (oracle/qarv_oracle.py) against golden vectors produced by the REFERENCE's own
classes (tests/golden/make_golden.py, imported from /root/reference in the build container).

Floating-point tolerance: the oracle issues the same PyTorch CPU ops as the reference, so it is
bit-identical on the generating machine; 2e-5 abs / 1e-5 rel leaves room for a different host CPU
(other SIMD width => other summation order).  Integer outputs (indexes, symbols, bitstreams) are
required exact up to a tiny flip budget for the same reason (round() of a value within 1e-6 of .5).
"""
import json
import os

import numpy as np
import pytest
import torch

import seeded_init
from oracle import qarv_oracle
from oracle import compressai_semantics as cs

# The goldens were generated in the build container (where /root/reference exists) from the imported reference; the oracle issues the same
# PyTorch CPU ops, so THERE it must reproduce them exactly -- 0 flips, byte-identical containers.  On another host
# (other CPU, other SIMD width => other summation order) a round() of a value within 1e-6 of .5 may fall the other way: small budget.
EXACT = os.path.isdir('/root/reference') or os.environ.get('LVAE_ORACLE_EXACT') == '1'
FLIP_BUDGET = 0.0 if EXACT else 2e-3


@pytest.fixture(scope='module')
def oracle_model(qarv_seeded_sd):
    m = qarv_oracle.QarvOracle(qarv_seeded_sd)
    m.compress_mode()
    return m


def _img(h, w, seed):
    u8 = seeded_init.synthetic_image_u8(h, w, seed)
    return torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0)


def test_param_inventory(qarv_seeded_sd):
    # 93.433 M parameters
    n = sum(v.size for v in qarv_seeded_sd.values())
    assert n == 93433400 or abs(n / 1e6 - 93.4334) < 1e-4


def test_pack_byte_strings_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_byte_strings.npz'))
    for i in range(3):
        lengths = g[f'case{i}.lengths'].tolist()
        joined = g[f'case{i}.joined'].tobytes()
        parts, o = [], 0
        for L in lengths:
            parts.append(joined[o:o + L]); o += L
        packed = qarv_oracle.pack_byte_strings(parts)
        assert packed == g[f'case{i}.packed'].tobytes()
        assert qarv_oracle.unpack_byte_string(packed) == parts


def test_tables_match_reference(golden_dir, oracle_model):
    g = np.load(os.path.join(golden_dir, 'discretized_gaussian_tables.npz'))
    dg = oracle_model.dg
    np.testing.assert_array_equal(dg.scale_table.numpy(), g['scale_table'])
    np.testing.assert_array_equal(dg._quantized_cdf.numpy(), g['quantized_cdf'])
    np.testing.assert_array_equal(dg._cdf_length.numpy(), g['cdf_length'])
    np.testing.assert_array_equal(dg._offset.numpy(), g['offset'])
    # int32[64,249], max length 247 (+2)
    assert g['quantized_cdf'].shape == (64, 249) and int(g['cdf_length'].max()) == 249
    # CDF post-conditions
    for i in range(64):
        L = int(g['cdf_length'][i])
        row = g['quantized_cdf'][i, :L]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) >= 1)


def test_cnx_block(golden_dir):
    g = np.load(os.path.join(golden_dir, 'cnx_block.npz'))
    for tag in ('c128k7', 'c192k7', 'c512k3', 'c384k5', 'c256k7', 'c512k1'):
        dim, k, mlp1000, h, w = g[f'{tag}.cfg'].tolist()
        shapes = qarv_oracle.cnx_shapes('blk', dim, k, mlp1000 / 1000.0)
        sd = {n: torch.from_numpy(seeded_init.seeded_tensor(f'blk.{tag}.{n[4:]}', s, 0)) for n, s in shapes}
        y = qarv_oracle.cnx_adaln(sd, 'blk', torch.from_numpy(g[f'{tag}.x']), torch.from_numpy(g[f'{tag}.emb']))
        np.testing.assert_allclose(y.numpy(), g[f'{tag}.y'], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('tag,seed', [('64x64', 0), ('128x192', 1), ('512x768', 7)])      # 512x768: the size the metric is quoted on
def test_full_model(golden_dir, oracle_model, tag, seed):
    if tag == '512x768' and not EXACT:
        pytest.skip('free-running full-size comparison: exact on the host that generated the golden; elsewhere one rounding flip cascades '
                    '(the -m gpu tests hold the HIP path to this golden teacher-forced)')
    g = np.load(os.path.join(golden_dir, f'qarv_base_{tag}.npz'))
    h, w = g['hw'].tolist()
    im = _img(h, w, seed)
    for lmb in g['lmbs'].tolist():
        key = f'lmb{int(lmb)}'
        tr = oracle_model.encode_trace(im, lmb, code=True)
        np.testing.assert_allclose(tr['emb'].numpy(), g[f'{key}.emb'], rtol=1e-5, atol=1e-6)
        for name, v in tr['enc_features'].items():
            a = v.numpy()
            ref = g[f'{key}.{name}']
            if a.shape != ref.shape:
                a = a[:, ::4, ::2, ::2]
            np.testing.assert_allclose(a, ref, rtol=1e-4, atol=2e-4)
        nsym = flips = iflips = 0
        for bi, blk in enumerate(tr['blocks']):
            np.testing.assert_allclose(blk['pm'].numpy(), g[f'{key}.b{bi}.pm'], rtol=1e-4, atol=2e-4)
            np.testing.assert_allclose(blk['pv'].numpy(), g[f'{key}.b{bi}.pv'], rtol=1e-4, atol=2e-4)
            sym, idx = blk['symbols'].numpy(), blk['indexes'].numpy()
            nsym += sym.size
            flips += int((sym != g[f'{key}.b{bi}.symbols']).sum())
            iflips += int((idx != g[f'{key}.b{bi}.indexes']).sum())
        assert flips <= FLIP_BUDGET * nsym and iflips <= FLIP_BUDGET * nsym, (flips, iflips, nsym)
        string = oracle_model.compress(im, lmb)
        if flips == 0 and iflips == 0:
            assert string == g[f'{key}.bitstream'].tobytes()
        xhat = oracle_model.decompress(string)
        if EXACT:
            assert flips == 0 and iflips == 0 and string == g[f'{key}.bitstream'].tobytes()
        np.testing.assert_allclose(xhat.numpy(), g[f'{key}.xhat'], rtol=0, atol=2e-6 if EXACT else (1e-4 if flips == 0 else 5e-2))
        assert float(g[f'{key}.xhat_from_z_maxdiff']) == 0.0
        zs = [b['z'] for b in tr['blocks']]
        x2 = oracle_model.decode_from_latents(lmb, zs)
        assert float((x2 - xhat).abs().max()) <= 1e-6


def test_progressive_decoding_t0(golden_dir, oracle_model):
    """scripts/qarv/robust-decoding.py:38-56 on the reference: conditional_sample with the first k+1 latents given and the
    others at their prior means (t = 0), and unconditional_sample at t = 0."""
    g = np.load(os.path.join(golden_dir, 'qarv_base_64x128_progressive.npz'))
    lmb, (h, w) = float(g['lmb']), g['hw']
    zs = [torch.from_numpy(g[f'z{i}']) for i in range(9)]
    for anchor in (0, 3, 8):
        lat = [z if i <= anchor else None for i, z in enumerate(zs)]
        x = oracle_model.decode_from_latents(lmb, lat, bhw_repeat=(1, h // 64, w // 64))
        np.testing.assert_allclose(x.numpy(), g[f'x{anchor}'], rtol=0, atol=2e-5)
    x = oracle_model.decode_from_latents(lmb, [None] * 9, bhw_repeat=(1, h // 64, w // 64))
    np.testing.assert_allclose(x.numpy(), g['x_uncond_t0'], rtol=0, atol=2e-5)


def _robust_cases(g):
    """latent lists of tests/golden/qarv_base_64x64_robust.npz (make_golden.py::golden_robust)."""
    zs = [torch.from_numpy(g[f'z{i}']) for i in range(9)]
    cases = {}
    for a in (0, 4, 8):
        cases[f'exclude{a}'] = [None if i == a else z for i, z in enumerate(zs)]
    for a in (2, 6):
        cases[f'reverse{a}'] = [None if i < a else z for i, z in enumerate(zs)]
    for a in (0, 3, 8):
        cases[f'single{a}'] = [z if i == a else None for i, z in enumerate(zs)]
    ed = list(zs)
    ed[1], ed[5] = torch.from_numpy(g['edited.z1']), torch.from_numpy(g['edited.z5'])
    cases['edited'] = ed
    return cases


def test_robust_decoding_uses_latents_verbatim(golden_dir, oracle_model):
    """'exclude' / 'reverse' / 'single' decodings of scripts/qarv/robust-decoding.py:44-49 and an edited latent: a supplied latent
    is used as it is (qarv/model.py:101-103) although its block's prior mean has changed."""
    g = np.load(os.path.join(golden_dir, 'qarv_base_64x64_robust.npz'))
    lmb, (h, w) = float(g['lmb']), g['hw']
    for name, lat in _robust_cases(g).items():
        x = oracle_model.decode_from_latents(lmb, lat, bhw_repeat=(1, h // 64, w // 64))
        np.testing.assert_allclose(x.numpy(), g[f'x.{name}'], rtol=0, atol=2e-5, err_msg=name)


def test_imcoding_evaluate_contract(golden_dir, oracle_model, tmp_path):
    """lvae/evaluation.py:15-67 semantics: bpp over ORIGINAL pixels incl. 4-byte (h,w) header; PSNR on the
    un-rounded float reconstruction; mean of per-image values; ragged sizes padded (coding.py:73-91)."""
    import math, struct
    with open(os.path.join(golden_dir, 'imcoding_evaluate.json')) as f:
        G = json.load(f)
    for key, ref in G['results'].items():
        lmb = float(key[3:])
        acc = {'bpp': [], 'mse': [], 'psnr': []}
        for (h, w), seed in zip(G['sizes'], G['seeds']):
            u8 = seeded_init.synthetic_image_u8(h, w, seed)
            pad = qarv_oracle.pad_divisible_by_u8(u8, 64)
            im = torch.from_numpy(pad).permute(2, 0, 1).float().div(255).unsqueeze(0)
            body = oracle_model.compress(im, lmb)
            blob = struct.pack('2H', h, w) + body                      # qarv/model.py:567-570
            xhat = oracle_model.decompress(blob[4:])[:, :, :h, :w].squeeze(0)
            real = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
            mse = (real - xhat).square().mean().item()
            acc['bpp'].append(len(blob) * 8 / float(h * w))
            acc['mse'].append(mse)
            acc['psnr'].append(-10 * math.log10(mse))
        for k in acc:
            assert abs(np.mean(acc[k]) - ref[k]) <= 2e-3 * abs(ref[k]) + 1e-6, (key, k, np.mean(acc[k]), ref[k])
