"""qres34m: oracle vs the reference's goldens (not-gpu) and the HIP path vs both (gpu)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

import seeded_init
from oracle import qres_oracle

# exact in the build container, where the goldens were generated (see tests/test_oracle_golden.py)
EXACT = os.path.isdir('/root/reference') or os.environ.get('LVAE_ORACLE_EXACT') == '1'
FLIP_BUDGET = 0.0 if EXACT else 2e-3


@pytest.fixture(scope='module')
def qres_sd():
    return seeded_init.seeded_state_dict(qres_oracle.qres_param_shapes(qres_oracle.qres34m_arch()), seed=0)


@pytest.fixture(scope='module')
def oracle(qres_sd):
    o = qres_oracle.QresOracle(qres_sd)
    o.compress_mode()
    return o


def _img(h, w, seed):
    u8 = seeded_init.synthetic_image_u8(h, w, seed)
    return torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0)


def test_inventory(qres_sd, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, 'qres34m_state_keys.json')))
    params = {k: v for k, v in ref.items() if '.discrete_gaussian.' not in k}
    assert {k: list(v.shape) for k, v in qres_sd.items()} == params
    assert sum(v.size for v in qres_sd.values()) == 34036602           # 34.037 M


def test_product_state_dict_and_tables(golden_dir):
    import lvae
    m = lvae.get_model('qres34m', lmb=64)
    ref = json.load(open(os.path.join(golden_dir, 'qres34m_state_keys.json')))
    assert set(m.state_dict().keys()) == set(ref.keys())
    assert not hasattr(m, 'default_lmb')                                  # eval-var-rate.py:41 relies on this for fixed-rate models
    m.compress_mode()
    g = np.load(os.path.join(golden_dir, 'gaussian_conditional_tables.npz'))
    dg = m._dg()
    assert np.array_equal(dg._quantized_cdf.numpy(), g['quantized_cdf']) and np.array_equal(dg._offset.numpy(), g['offset'])
    assert np.array_equal(dg._cdf_length.numpy(), g['cdf_length']) and np.array_equal(dg.scale_table.numpy(), g['scale_table'])
    # a checkpoint saved before compress_mode() (empty buffers) or without some entropy-model buffers must load
    sd = {k: v for k, v in lvae.get_model('qres34m').state_dict().items() if not k.endswith('scale_bound')}
    m.load_state_dict(sd)


@pytest.mark.parametrize('tag,seed', [('64x64', 0), ('128x192', 1), ('512x768', 7)])      # 512x768: BASELINE config 3's size
def test_oracle_matches_reference(golden_dir, oracle, tag, seed):
    if tag == '512x768' and not EXACT:
        pytest.skip('free-running full-size comparison: exact on the host that generated the golden; elsewhere one rounding flip cascades '
                    '(the -m gpu tests hold the HIP path to this golden teacher-forced)')
    g = np.load(os.path.join(golden_dir, f'qres34m_{tag}.npz'))
    h, w = g['hw'].tolist()
    im = _img(h, w, seed)
    tr = oracle.encode_trace(im, code=True)
    assert tuple(g['smallest'].tolist()) == tr['smallest']
    n = flips = 0
    for bi, blk in enumerate(tr['blocks']):
        np.testing.assert_allclose(blk['pm'].numpy(), g[f'b{bi}.pm'], rtol=1e-4, atol=2e-4)
        n += blk['symbols'].numel()
        flips += int((blk['symbols'].numpy() != g[f'b{bi}.symbols']).sum()) + int((blk['indexes'].numpy() != g[f'b{bi}.indexes']).sum())
        if flips == 0:
            assert blk['strings'][0] == g[f'b{bi}.string'].tobytes()
    assert flips <= FLIP_BUDGET * n
    obj = oracle.compress(im)
    assert len(pickle.dumps(obj + [(h, w)])) == int(g['pickle_bytes']) or flips
    xhat = oracle.decompress(obj)
    np.testing.assert_allclose(xhat.numpy(), g['xhat'], rtol=0, atol=2e-6 if EXACT else (1e-4 if flips == 0 else 5e-2))


# Absolute per-case ceilings on the MI355X: no symbol flip, at most two scale indexes (qres34m: 0-2 observed).
MAX_SYM_FLIPS, MAX_IDX_FLIPS = 0, 2


def _hip_golden_case(product, g, im, case, n_blocks=12):
    """HIP path vs one reference golden: per-block symbols / scale indexes under absolute ceilings, every block's rANS stream
    byte-identical (for a block with a flipped scale index: after substituting the reference's value, through the same coder),
    reconstruction within 1e-4 -- never waived.  One row goes to the parity report (tests/conftest.py)."""
    from conftest import parity_record
    from lvae.models.entropy_coding import rans_encode_streams
    tr = product.encode_trace(im)
    obj = product.compress(im)
    assert len(tr) == n_blocks and len(obj) == n_blocks + 1 and tuple(obj[-1]) == tuple(g['smallest'].tolist())
    tables = product._dg().host_tables()
    n = flips = iflips = 0
    same = True
    for bi, blk in enumerate(tr):
        gs, gi = g[f'b{bi}.symbols'].reshape(-1), g[f'b{bi}.indexes'].reshape(-1)
        f, fi = int((blk['symbols'].reshape(-1) != gs).sum()), int((blk['indexes'].reshape(-1) != gi).sum())
        n += gs.size; flips += f; iflips += fi
        gold = g[f'b{bi}.string'].tobytes()
        if f == 0 and fi == 0:
            assert obj[bi][0] == gold, f'{case}: block {bi} stream differs although its symbols and indexes match'
        else:
            same = False
            fixed = rans_encode_streams(tables, [np.ascontiguousarray(gs.astype(np.int32))], [np.ascontiguousarray(gi.astype(np.uint8))], 1)[0]
            assert fixed == gold, f'{case}: block {bi}'
    xhat = product.decompress(obj)
    err = float((xhat.cpu() - torch.from_numpy(g['xhat'])).abs().max())
    # every flip must be a guard-band event and every element within rounding noise of the reference (tests/parity_util.py):
    # teacher-forced with the reference's latents, so a flip is never the cascade of an earlier one
    import parity_util
    gb = []
    for bi in range(n_blocks):
        b = {k: g[f'b{bi}.{k}'] for k in ('pm', 'pv', 'qm', 'indexes', 'symbols')}
        b['z'] = torch.from_numpy(b['symbols'].astype(np.float32) + b['pm'].astype(np.float32))
        gb.append(b)
    guard = parity_util.check_blocks(case, product.encode_trace(im, full=True, force_z=[b['z'] for b in gb]), gb,
                                     product._dg().scale_table.cpu().numpy(), product._packed.scale_bound)
    assert guard['n'] == n
    parity_record(case, flips, iflips, n, err, same, guard)
    assert np.array_equal(tr[0]['symbols'].reshape(-1), g['b0.symbols'].reshape(-1))
    assert flips <= MAX_SYM_FLIPS and iflips <= MAX_IDX_FLIPS, (case, flips, iflips, n)
    assert err <= 1e-4, (case, err)
    return flips + iflips


@pytest.fixture(scope='module')
def product(qres_sd):
    import lvae
    m = lvae.get_model('qres34m')
    full = m.state_dict()
    for k, v in qres_sd.items():
        full[k] = torch.from_numpy(v)
    m.load_state_dict(full)
    m.compress_mode()                      # eval-fix-rate.py order: tables on the CPU, then .to(device)
    m = m.to('cuda:0').eval()
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('tag,seed', [('64x64', 0), ('128x192', 1)])
def test_hip_matches_reference(golden_dir, product, tag, seed):
    g = np.load(os.path.join(golden_dir, f'qres34m_{tag}.npz'))
    h, w = g['hw'].tolist()
    _hip_golden_case(product, g, _img(h, w, seed).cuda(), f'qres34m {tag}')


@pytest.mark.gpu
def test_hip_file_round_trip_and_batch(product, tmp_path):
    from PIL import Image
    u8 = seeded_init.synthetic_image_u8(100, 70, 11)
    Image.fromarray(u8).save(tmp_path / 'a.png')
    product.compress_file(tmp_path / 'a.png', tmp_path / 'a.bits')
    obj = pickle.load(open(tmp_path / 'a.bits', 'rb'))
    assert obj[-1] == (100, 70) and obj[-2] == (1, 384, 2, 2) and len(obj) == 14
    x = product.decompress_file(tmp_path / 'a.bits')
    assert x.shape == (1, 3, 100, 70)
    ims = torch.cat([_img(128, 64, s) for s in (1, 2, 3, 4, 5)], 0).cuda()
    objs = product.compress_batch(ims)
    for i in range(5):
        assert objs[i] == product.compress(ims[i:i + 1])
    xb = product.decompress_batch(objs)
    assert torch.equal(xb[2:3], product.decompress(objs[2]))


@pytest.mark.gpu
def test_large_group_of_small_images_equals_singles(product):
    """qres34m's width-384 blocks (hidden 768) sit on maps of 32x32 ... 4x4 for a 256x256 image -- fewer than 1536
    rows per image, i.e. split-K layers.  A pipeline group of 48 such images has M = 49152 rows at stride 8, the fused-MLP row
    threshold of that block shape; the fused kernel has the S = 1 summation order, the single-image path the split-K one, so the rule
    must refuse the fused form there whatever the batch: the strings of the 48-image group equal the single-image strings, and
    the group decodes to the single-image reconstructions."""
    m = product
    groups = m.pipeline_groups
    ims = torch.cat([_img(256, 256, 400 + i) for i in range(48)], 0).cuda()
    try:
        m.pipeline_groups = 1
        objs = m.compress_batch(ims)
        xb = m.decompress_batch(objs)
    finally:
        m.pipeline_groups = groups
    pl = next(p for k, p in m._plans.items() if k[0] == 'enc' and k[1] == 48)
    from lvae import _native
    import ctypes
    for fn, a, lab, _s in pl.ops:
        if callable(fn) and getattr(fn, 'lvae_name', '') == 'lvae_mlp_h2f':
            d = ctypes.cast(a[0], ctypes.POINTER(_native.MlpDesc)).contents
            assert (d.C, d.hid) != (384, 768), lab                       # only the stride-4 blocks (192 / 384) are fused here
    for i in (0, 17, 47):
        assert objs[i] == m.compress(ims[i:i + 1]), i
        assert torch.equal(xb[i:i + 1], m.decompress(objs[i])), i


# ----------------------------------------------------------------------------------- qres34m_lossless
@pytest.fixture(scope='module')
def lossless_sd():
    return seeded_init.seeded_state_dict(qres_oracle.qres_param_shapes(qres_oracle.qres34m_lossless_arch()), seed=0)


def test_lossless_inventory(lossless_sd, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, 'qres34m_lossless_state_keys.json')))
    assert {k: list(v.shape) for k, v in lossless_sd.items()} == ref
    import lvae
    m = lvae.get_model('qres34m_lossless')
    own = {k: list(v.shape) for k, v in m.state_dict().items() if 'discrete_gaussian' not in k}
    assert own == ref


def test_lossless_oracle_matches_reference(golden_dir, lossless_sd):
    """GaussianNLLOutputNet.compress/decompress (qresvae/model.py:69-94): per-pixel means / scale indexes / symbols and the
    final string against the reference; the decode is bit-exact (the reference's evaluate-lossless.py assertion)."""
    g = np.load(os.path.join(golden_dir, 'qres34m_lossless_64x128.npz'))
    h, w = g['hw'].tolist()
    o = qres_oracle.QresOracle(lossless_sd, arch=qres_oracle.qres34m_lossless_arch())
    o.compress_mode()
    np.testing.assert_allclose(o.out_dg.scale_table.numpy(), g['scale_table'], rtol=1e-6)
    u8 = seeded_init.synthetic_image_u8(h, w, int(g['img_seed']))
    im = torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0)
    tr = {}
    obj = o.compress(im, trace=tr)
    assert len(obj) == 14 and tuple(obj[-2]) == tuple(g['smallest'].tolist())
    lat_same = all(obj[i][0] == g[f'string{i}'].tobytes() for i in range(12))
    n = tr['symbols'].numel()
    flips = int((tr['symbols'].numpy() != g['out.symbols']).sum()) + int((tr['indexes'].numpy() != g['out.indexes']).sum())
    # (exact in the container that made the golden; another host's torch CPU kernels may sum a conv in another order and move one
    #  of the 24 576 per-pixel elements across a rounding boundary: seen on the MI355X box's host)
    assert flips <= (2 if lat_same else 0.05 * n), (flips, n)
    if lat_same and flips == 0:
        assert obj[-1][0] == g['out.string'].tobytes()
        assert len(pickle.dumps(obj + [(h, w)])) == int(g['pickle_bytes'])
    xhat = o.decompress(obj)
    assert np.array_equal(torch.round(xhat * 255.0).to(torch.uint8)[0].permute(1, 2, 0).numpy(), u8)      # lossless
    assert bool(g['lossless'])


@pytest.fixture(scope='module')
def lossless_product(lossless_sd):
    import lvae
    m = lvae.get_model('qres34m_lossless')
    full = m.state_dict()
    for k, v in lossless_sd.items():
        full[k] = torch.from_numpy(v)
    m.load_state_dict(full)
    m.compress_mode()
    return m.to('cuda:0').eval()


def _lossless_pixel_guard(pl, g, sym, idx, h, w):
    """Guard-band proof for the per-pixel stream of GaussianNLLOutputNet (qresvae/model.py:69-94) against the reference golden.  With the
    12 latent strings byte-identical the output net sees the reference's feature up to rounding noise, so: every raw mean / log-scale
    (the conv_mean | conv_scale outputs, `out.raw_*` of the golden) within VAL_ATOL / LNS_TOL for ALL 3*H*W elements; a flipped SYMBOL
    is a flipped rounded mean -- pm = round(m * 127.5 + 127.5) differs by one because m * 127.5 + 127.5 sits on a half-integer (margin
    <= 127.5 * VAL_ATOL on both sides), the image sample being the same integer on both sides; a flipped INDEX has its scale
    exp(ls) / bin within IDX_BAND of the table threshold between the two rows on both sides.  -> the dict check_blocks returns."""
    import parity_util as pu
    raw = pl.px_raw.cpu().numpy().reshape(1, h, w, 6).astype(np.float64)                  # NHWC: mean c0..2 | log-scale c0..2
    m_hip, ls_hip = raw[..., :3].transpose(0, 3, 1, 2), raw[..., 3:].transpose(0, 3, 1, 2)
    m_ref, ls_ref = g['out.raw_mean'].astype(np.float64), g['out.raw_logscale'].astype(np.float64)
    st = dict(sym_flips=0, idx_flips=0, n=sym.size, max_dval=float(np.abs(m_hip - m_ref).max()), max_dlns=float(np.abs(ls_hip - ls_ref).max()),
              worst_sym_margin=0.0, worst_idx_margin=0.0)
    assert st['max_dval'] <= pu.VAL_ATOL and st['max_dlns'] <= pu.LNS_TOL, st
    table = g['scale_table'].astype(np.float64)
    lnbin = -4.848116360536466                                                            # math.log(1 / 127.5)
    for pos in zip(*np.nonzero(idx != g['out.indexes'])):
        ia, ib = int(idx[pos]), int(g['out.indexes'][pos])
        assert abs(ia - ib) == 1, (pos, ia, ib)
        thr = table[min(ia, ib)]
        mg = max(abs(max(np.exp(ls_hip[pos] - lnbin), 0.11) / thr - 1), abs(max(np.exp(ls_ref[pos] - lnbin), 0.11) / thr - 1))
        st['idx_flips'] += 1
        st['worst_idx_margin'] = max(st['worst_idx_margin'], float(mg))
        assert mg <= pu.IDX_BAND, (pos, mg)
    for pos in zip(*np.nonzero(sym != g['out.symbols'])):
        sa, sb = int(sym[pos]), int(g['out.symbols'][pos])
        assert abs(sa - sb) == 1, (pos, sa, sb)
        va, vb = m_hip[pos] * 127.5 + 127.5, m_ref[pos] * 127.5 + 127.5                      # the value torch.round sees (:72), exactly
        # ... and as both sides compute it: two fp32 operations, then round-half-even (a value that lands ON the half-integer in fp32 goes to
        # the even neighbour, whatever its exact value was)
        f32 = np.float32
        ra = np.rint(f32(f32(f32(m_hip[pos]) * f32(127.5)) + f32(127.5)))
        rb = np.rint(f32(f32(f32(m_ref[pos]) * f32(127.5)) + f32(127.5)))
        assert abs(float(ra) - float(rb)) == 1, (pos, va, vb, ra, rb)                         # the two rounded means are neighbours
        half = (float(ra) + float(rb)) / 2.0
        mg = max(abs(va - half), abs(vb - half))
        st['sym_flips'] += 1
        st['worst_sym_margin'] = max(st['worst_sym_margin'], float(mg))
        assert mg <= 127.5 * pu.VAL_ATOL, (pos, va, vb, mg)
    return st


@pytest.mark.gpu
def test_lossless_hip_matches_reference_and_is_lossless(golden_dir, lossless_product, tmp_path):
    from PIL import Image
    m = lossless_product
    g = np.load(os.path.join(golden_dir, 'qres34m_lossless_64x128.npz'))
    h, w = g['hw'].tolist()
    u8 = seeded_init.synthetic_image_u8(h, w, int(g['img_seed']))
    im = torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0).cuda()
    obj = m.compress(im)
    assert len(obj) == 14 and tuple(obj[-2]) == tuple(g['smallest'].tolist())
    lat_same = all(obj[i][0] == g[f'string{i}'].tobytes() for i in range(12))
    pl = m._plan('enc', 1, h, w)
    sym, idx, pm = pl.px_sym.cpu().numpy().reshape(1, 3, h, w), pl.px_idx.cpu().numpy().reshape(1, 3, h, w), pl.px_pm.cpu().numpy().reshape(1, 3, h, w)
    n = sym.size
    flips = int((sym != g['out.symbols']).sum()) + int((idx != g['out.indexes']).sum())
    print(f'qres34m_lossless: latent strings identical: {lat_same}; pixel stream flips {flips} of {n}; max|dpm| {np.abs(pm - g["out.pm"]).max()}')
    guard = _lossless_pixel_guard(pl, g, sym, idx, h, w)
    from conftest import parity_record
    parity_record('qres34m_lossless 64x128 (12 latent streams + per-pixel stream)', int((sym != g['out.symbols']).sum()),
                  int((idx != g['out.indexes']).sum()), n, 0.0, lat_same and flips == 0, guard)
    # absolute ceilings (MI355X today: latent strings byte-identical, 2 of 24 576 per-pixel elements off by one table row / unit)
    assert lat_same and flips <= 4, (lat_same, flips, n)
    if lat_same and flips == 0:
        assert obj[-1][0] == g['out.string'].tobytes()
        assert len(pickle.dumps(obj + [(h, w)])) == int(g['pickle_bytes'])
    xhat = m.decompress(obj)
    assert np.array_equal(torch.round(xhat * 255.0).to(torch.uint8)[0].permute(1, 2, 0).cpu().numpy(), u8)           # lossless
    # files + batches (scripts/qresvae/evaluate-lossless.py:21-30), ragged size -> padded, cropped back, still exact
    u2 = seeded_init.synthetic_image_u8(100, 70, 21)
    Image.fromarray(u2).save(tmp_path / 'a.png')
    m.compress_file(tmp_path / 'a.png', tmp_path / 'a.bits')
    fake = m.decompress_file(tmp_path / 'a.bits').squeeze(0).cpu()
    assert np.array_equal(torch.round(fake * 255.0).to(torch.uint8).permute(1, 2, 0).numpy(), u2)
    ims = torch.cat([torch.from_numpy(seeded_init.synthetic_image_u8(64, 128, s)).permute(2, 0, 1).float().div(255).unsqueeze(0)
                     for s in (31, 32, 33)], 0).cuda()
    objs = m.compress_batch(ims)
    assert objs[1] == m.compress(ims[1:2])
    xb = m.decompress_batch(objs)
    assert torch.equal(torch.round(xb * 255.0), torch.round(ims * 255.0))


# ----------------------------------------------------------------------------------- qres17m
@pytest.fixture(scope='module')
def q17_sd():
    return seeded_init.seeded_state_dict(qres_oracle.qres_param_shapes(qres_oracle.qres17m_arch()), seed=0)


def test_qres17m_inventory(q17_sd, golden_dir):
    ref = json.load(open(os.path.join(golden_dir, 'qres17m_state_keys.json')))
    assert {k: list(v.shape) for k, v in q17_sd.items()} == ref
    import lvae
    m = lvae.get_model('qres17m')
    assert {k: list(v.shape) for k, v in m.state_dict().items() if 'discrete_gaussian' not in k} == ref
    assert sum(v.size for v in q17_sd.values()) == 16678842


def test_qres17m_oracle_matches_reference(golden_dir, q17_sd):
    """4x4/s4 patch down-sampling, nearest x4 Upsample and the two stride-2 transposed convs (zoo.py:121-166)."""
    g = np.load(os.path.join(golden_dir, 'qres17m_64x128.npz'))
    h, w = g['hw'].tolist()
    o = qres_oracle.QresOracle(q17_sd, arch=qres_oracle.qres17m_arch())
    o.compress_mode()
    im = _img(h, w, int(g['img_seed']))
    tr = o.encode_trace(im, code=True)
    assert tuple(g['smallest'].tolist()) == tr['smallest'] and len(tr['blocks']) == 12
    n = flips = 0
    for bi, blk in enumerate(tr['blocks']):
        n += blk['symbols'].numel()
        flips += int((blk['symbols'].numpy() != g[f'b{bi}.symbols']).sum()) + int((blk['indexes'].numpy() != g[f'b{bi}.indexes']).sum())
        if flips == 0:
            assert blk['strings'][0] == g[f'b{bi}.string'].tobytes()
    assert flips <= FLIP_BUDGET * n
    obj = o.compress(im)
    xhat = o.decompress(obj)
    np.testing.assert_allclose(xhat.numpy(), g['xhat'], rtol=0, atol=2e-6 if EXACT else (1e-4 if flips == 0 else 5e-2))


@pytest.fixture(scope='module')
def q17_product(q17_sd):
    import lvae
    m = lvae.get_model('qres17m')
    full = m.state_dict()
    for k, v in q17_sd.items():
        full[k] = torch.from_numpy(v)
    m.load_state_dict(full)
    m.compress_mode()
    return m.to('cuda:0').eval()


@pytest.mark.gpu
def test_qres17m_hip_matches_reference(golden_dir, q17_product):
    m = q17_product
    g = np.load(os.path.join(golden_dir, 'qres17m_64x128.npz'))
    h, w = g['hw'].tolist()
    im = _img(h, w, int(g['img_seed'])).cuda()
    _hip_golden_case(m, g, im, 'qres17m 64x128')
    ims = torch.cat([_img(128, 64, s) for s in (1, 2, 3)], 0).cuda()
    objs = m.compress_batch(ims)
    assert objs[2] == m.compress(ims[2:3])
    assert torch.equal(m.decompress_batch(objs)[1:2], m.decompress(objs[1]))
