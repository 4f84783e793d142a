"""tests/golden/make_golden.py -- generates the committed golden fixtures (build container only).

Imports the REFERENCE's own Python from /root/reference under the stand-in modules of ref_shims.py,
instantiates its classes (`lvae.get_model('qarv_base')`, `ConvNeXtBlockAdaLN`, `imcoding_evaluate`,
`pack_byte_strings` ...), loads seeded synthetic weights (lossy-vae_amd/seeded_init.py) and records
inputs/outputs as small .npz files next to this script.  The fixtures are DATA; no reference source
text is stored.  Run:  python tests/golden/make_golden.py
"""
import importlib.util
import io
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get('LVAE_GOLDEN_OUT', HERE)          # where the fixtures are written (default: next to this script)
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
spec = importlib.util.spec_from_file_location('seeded_init', os.path.join(REPO, 'lossy-vae_amd', 'seeded_init.py'))
seeded_init = importlib.util.module_from_spec(spec)
spec.loader.exec_module(seeded_init)

import lvae  # noqa: E402  (the reference package)
import lvae.models.common as ref_common  # noqa: E402
import lvae.utils.coding as ref_coding  # noqa: E402
from lvae.evaluation import imcoding_evaluate  # noqa: E402

assert lvae.__file__.startswith('/root/reference'), lvae.__file__
torch.manual_seed(0)
torch.set_num_threads(8)


def load_seeded(model, seed=0):
    sd = model.state_dict()
    new = {}
    for k, v in sd.items():
        a = seeded_init.seeded_tensor(k, tuple(v.shape), seed)
        new[k] = v if a is None else torch.from_numpy(a)
    model.load_state_dict(new)
    return model


def npf(t):
    return t.detach().cpu().numpy()


def image_tensor(h, w, seed=0, kind='natural'):
    u8 = seeded_init.synthetic_image_u8(h, w, seed, kind)
    return torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0), u8


@torch.no_grad()
def golden_cnx_block():
    """One ConvNeXtBlockAdaLN (common.py:110-161) per (dim, k, mlp_ratio) used on a small map."""
    out = {}
    for tag, (dim, k, mlp, h, w) in {'c128k7': (128, 7, 1.5, 9, 11), 'c192k7': (192, 7, 2, 8, 8),
                                     'c512k3': (512, 3, 3, 4, 6), 'c384k5': (384, 5, 2, 6, 5),
                                     'c256k7': (256, 7, 1.75, 8, 7), 'c512k1': (512, 1, 4, 2, 3)}.items():
        blk = ref_common.ConvNeXtBlockAdaLN(dim, 256, kernel_size=k, mlp_ratio=mlp).eval()
        sd = {kk: torch.from_numpy(seeded_init.seeded_tensor(f'blk.{tag}.{kk}', tuple(v.shape), 0))
              for kk, v in blk.state_dict().items()}
        blk.load_state_dict(sd)
        g = np.random.Generator(np.random.Philox(key=1234 + dim + k))
        x = torch.from_numpy(g.normal(0, 1, size=(2, dim, h, w)).astype(np.float32))
        emb = torch.from_numpy(g.normal(0, 1, size=(2, 256)).astype(np.float32))
        y = blk(x, emb)
        out[f'{tag}.x'], out[f'{tag}.emb'], out[f'{tag}.y'] = npf(x), npf(emb), npf(y)
        out[f'{tag}.cfg'] = np.array([dim, k, int(mlp * 1000), h, w])
    np.savez_compressed(os.path.join(OUT, 'cnx_block.npz'), **out)
    print('cnx_block.npz', len(out))


@torch.no_grad()
def golden_qarv(model, h, w, lmbs, tag, img_seed=0, full_features=True):
    im, _ = image_tensor(h, w, img_seed)
    out = {'hw': np.array([h, w]), 'img_seed': np.array(img_seed), 'lmbs': np.array(lmbs, dtype=np.float64)}
    for lmb in lmbs:
        key = f'lmb{int(lmb)}'
        emb = model._get_lmb_embedding(lmb, n=1)
        out[f'{key}.emb'] = npf(emb)
        # encoder features (common.py:89-98)
        x = model.preprocess_input(im)
        _, feats = model.encoder(x, emb)
        for k, v in feats.items():
            a = npf(v)
            out[f'{key}.{k}'] = a if (full_features or a.size <= 65536) else a[:, ::4, ::2, ::2]
        # per-block stats: hook the VRLV blocks while the reference runs compress()
        recs = []
        hooks = []
        for blk in model.dec_blocks:
            if getattr(blk, 'is_latent_block', False):
                dg = blk.discrete_gaussian
                rec = {}
                recs.append(rec)

                def mk(rec, dg):
                    orig_bi, orig_c = dg.build_indexes, dg.compress

                    def bi(pv):
                        idx = orig_bi(pv)
                        rec['pv'], rec['indexes'] = npf(pv), npf(idx).astype(np.uint8)
                        return idx

                    def comp(qm, indexes, means=None):
                        rec['qm'], rec['pm'] = npf(qm), npf(means)
                        rec['symbols'] = npf(dg.quantize(qm, 'symbols', means)).astype(np.int32)
                        s = orig_c(qm, indexes, means=means)
                        rec['string'] = np.frombuffer(s[0], dtype=np.uint8)
                        return s
                    dg.build_indexes, dg.compress = bi, comp
                    return lambda: (setattr(dg, 'build_indexes', orig_bi), setattr(dg, 'compress', orig_c))
                hooks.append(mk(rec, dg))
        string = model.compress(im, lmb)
        for hk in hooks:
            hk()
        for bi, rec in enumerate(recs):
            for k, v in rec.items():
                out[f'{key}.b{bi}.{k}'] = v
        out[f'{key}.bitstream'] = np.frombuffer(string, dtype=np.uint8)
        xhat = model.decompress(string)
        out[f'{key}.xhat'] = npf(xhat)
        # coder-independent decoder check (qarv/model.py:365-395): conditional_sample with the true z
        zs = [torch.from_numpy(r['symbols']).float() + torch.from_numpy(r['pm']) for r in recs]
        xs = model.conditional_sample(lmb, latents=zs)
        out[f'{key}.xhat_from_z_maxdiff'] = np.array(float((xs - xhat).abs().max()))
        # estimated bits from the eval-mode likelihood (qarv/model.py:95-96), for the coder-size check
        model.eval()
        _, stats = model.forward_end2end(im, lmb=model.expand_to_tensor(lmb, n=1))
        bits = [float(st['kl'].sum()) / np.log(2) for st in stats]
        out[f'{key}.est_bits'] = np.array(bits)
        print(tag, key, 'bytes', len(string), 'est bits/8', sum(bits) / 8, 'sym range',
              [(int(r['symbols'].min()), int(r['symbols'].max())) for r in recs],
              'idx range', [(int(r['indexes'].min()), int(r['indexes'].max())) for r in recs],
              'dec-vs-sample', float(out[f'{key}.xhat_from_z_maxdiff']))
    np.savez_compressed(os.path.join(OUT, f'qarv_base_{tag}.npz'), **out)


@torch.no_grad()
def golden_tables(model):
    """DiscretizedGaussian.update() tables as produced by the reference subclass (entropy_coding.py:52-82)
    on top of the CompressAI-semantics stand-in."""
    dg = model.dec_blocks[0].discrete_gaussian
    np.savez_compressed(os.path.join(OUT, 'discretized_gaussian_tables.npz'),
                        scale_table=npf(dg.scale_table), quantized_cdf=npf(dg._quantized_cdf),
                        cdf_length=npf(dg._cdf_length), offset=npf(dg._offset))
    print('tables', tuple(dg._quantized_cdf.shape), int(dg._cdf_length.max()))


def golden_pack():
    """pack_byte_strings / unpack_byte_string (lvae/utils/coding.py:26-70) known-answer vectors."""
    cases = [[b'', b'\x01\x02\x03\x04'], [bytes(range(7)), b'', bytes(range(250, 256))], [b'\xff' * 300]]
    out = {}
    for i, c in enumerate(cases):
        packed = ref_coding.pack_byte_strings(c)
        assert ref_coding.unpack_byte_string(packed) == c
        out[f'case{i}.packed'] = np.frombuffer(packed, dtype=np.uint8)
        out[f'case{i}.lengths'] = np.array([len(s) for s in c])
        out[f'case{i}.joined'] = np.frombuffer(b''.join(c), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'pack_byte_strings.npz'), **out)


@torch.no_grad()
def golden_imcoding(model):
    """imcoding_evaluate (lvae/evaluation.py:15-67) on a 3-image synthetic folder of ragged sizes
    (exercises pad_divisible_by, coding.py:73-91, and the crop in decompress_file, qarv/model.py:581)."""
    from PIL import Image
    sizes = [(70, 100), (64, 64), (130, 65)]
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for i, (h, w) in enumerate(sizes):
            Image.fromarray(seeded_init.synthetic_image_u8(h, w, seed=100 + i)).save(Path(d) / f'im{i}.png')
        for lmb in (2048.0, 128.0):
            model.default_lmb = lmb
            r = imcoding_evaluate(model, d)
            res[f'lmb{int(lmb)}'] = r
        model.default_lmb = model.lmb_range[1]
    with open(os.path.join(OUT, 'imcoding_evaluate.json'), 'w') as f:
        json.dump({'sizes': sizes, 'seeds': [100, 101, 102], 'results': res}, f, indent=1)
    print('imcoding', res)


@torch.no_grad()
def golden_progressive(model, h, w, lmb, tag, img_seed=0):
    """scripts/qarv/robust-decoding.py:38-56: progressive decoding -- conditional_sample with the first k+1 latents of an encoded
    image given and the rest drawn from the prior at temperature t = 0 (i.e. z = prior mean: deterministic)."""
    im, _ = image_tensor(h, w, img_seed)
    model.eval()
    _, stats_all = model.forward_end2end(im, lmb=model.expand_to_tensor(lmb, n=1), get_latent=True)
    L = len(stats_all)
    out = {'hw': np.array([h, w]), 'img_seed': np.array(img_seed), 'lmb': np.array(float(lmb)),
           'bits': np.array([float(st['kl'].sum()) / np.log(2) for st in stats_all])}
    for i, st in enumerate(stats_all):
        out[f'z{i}'] = npf(st['z'])
    for anchor in range(L):
        latents = [st['z'] if i <= anchor else None for i, st in enumerate(stats_all)]
        x = model.conditional_sample(lmb=lmb, latents=latents, bhw_repeat=(1, h // 64, w // 64), t=0)
        out[f'x{anchor}'] = npf(x).astype(np.float32)
    x = model.unconditional_sample(lmb, bhw_repeat=(1, h // 64, w // 64), t=0)
    out['x_uncond_t0'] = npf(x).astype(np.float32)
    print('progressive', tag, 'psnr vs input:',
          [round(float(-10 * np.log10(np.mean((out[f'x{a}'] - npf(im)) ** 2))), 2) for a in range(L)])
    np.savez_compressed(os.path.join(OUT, f'qarv_base_{tag}_progressive.npz'), **out)


@torch.no_grad()
def golden_robust(model, h, w, lmb, tag, img_seed=0):
    """The other three decodings of scripts/qarv/robust-decoding.py:44-49 ('exclude', 'reverse', 'single') plus one EDITED latent:
    conditional_sample must use a supplied latent verbatim (qarv/model.py:101-103) even when the prior means of that block differ
    from the ones that produced it (an earlier block is missing / edited).  t = 0: missing latents = prior mean."""
    im, _ = image_tensor(h, w, img_seed)
    model.eval()
    _, stats_all = model.forward_end2end(im, lmb=model.expand_to_tensor(lmb, n=1), get_latent=True)
    L = len(stats_all)
    zs = [st['z'] for st in stats_all]
    out = {'hw': np.array([h, w]), 'img_seed': np.array(img_seed), 'lmb': np.array(float(lmb))}
    for i, z in enumerate(zs):
        out[f'z{i}'] = npf(z)
    bhw = (1, h // 64, w // 64)
    cases = {}
    for a in (0, 4, 8):
        cases[f'exclude{a}'] = [None if i == a else z for i, z in enumerate(zs)]
    for a in (2, 6):
        cases[f'reverse{a}'] = [None if i < a else z for i, z in enumerate(zs)]
    for a in (0, 3, 8):
        cases[f'single{a}'] = [z if i == a else None for i, z in enumerate(zs)]
    edited = [z.clone() for z in zs]
    edited[1] = edited[1] * 0.5 + 0.25            # off the integer grid of its prior mean
    edited[5] = torch.flip(edited[5], dims=[3]) * 0.75
    cases['edited'] = edited
    out['edited.z1'], out['edited.z5'] = npf(edited[1]), npf(edited[5])
    for name, lat in cases.items():
        x = model.conditional_sample(lmb=lmb, latents=lat, bhw_repeat=bhw, t=0)
        out[f'x.{name}'] = npf(x).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f'qarv_base_{tag}_robust.npz'), **out)
    print('robust', tag, sorted(cases))


@torch.no_grad()
def golden_qres(model, h, w, tag, img_seed=0, model_name='qres34m'):
    """qres34m (qresvae/model.py:649-725): per-block indexes/symbols/strings, reconstruction, pickle container size."""
    import pickle
    im, _ = image_tensor(h, w, img_seed)
    out = {'hw': np.array([h, w]), 'img_seed': np.array(img_seed)}
    recs, hooks = [], []
    for blk in model.decoder.dec_blocks:
        if hasattr(blk, 'discrete_gaussian'):
            dg = blk.discrete_gaussian
            rec = {}
            recs.append(rec)

            def mk(rec, dg):
                orig_bi, orig_c = dg.build_indexes, dg.compress

                def bi(pv):
                    idx = orig_bi(pv)
                    rec['pv'], rec['indexes'] = npf(pv), npf(idx).astype(np.uint8)
                    return idx

                def comp(qm, indexes, means=None):
                    rec['pm'], rec['qm'] = npf(means), npf(qm)
                    rec['symbols'] = npf(dg.quantize(qm, 'symbols', means)).astype(np.int32)
                    s = orig_c(qm, indexes, means=means)
                    rec['string'] = np.frombuffer(s[0], dtype=np.uint8)
                    return s
                dg.build_indexes, dg.compress = bi, comp
                return lambda: (setattr(dg, 'build_indexes', orig_bi), setattr(dg, 'compress', orig_c))
            hooks.append(mk(rec, dg))
    obj = model.compress(im)
    for hk in hooks:
        hk()
    for bi, rec in enumerate(recs):
        for k, v in rec.items():
            out[f'b{bi}.{k}'] = v
    out['smallest'] = np.array(obj[-1])
    out['pickle_bytes'] = np.array(len(pickle.dumps(obj + [(h, w)])))
    xhat = model.decompress(obj)
    out['xhat'] = npf(xhat)
    print(model_name, tag, 'payload bytes', sum(len(r['string']) for r in recs), 'pickle', int(out['pickle_bytes']),
          'sym range', [(int(r['symbols'].min()), int(r['symbols'].max())) for r in recs][:6],
          'idx range', [(int(r['indexes'].min()), int(r['indexes'].max())) for r in recs][:6])
    np.savez_compressed(os.path.join(OUT, f'{model_name}_{tag}.npz'), **out)


@torch.no_grad()
def golden_qres_lossless(model, h, w, tag, img_seed=0):
    """qres34m_lossless (qresvae/zoo.py:63-118, model.py:16-94,649-687): the 12 latent strings as for qres34m plus the final
    per-pixel string of GaussianNLLOutputNet (3*H*W symbols, 128-entry scale table); the decode must be bit-exact."""
    import pickle
    im, u8 = image_tensor(h, w, img_seed)
    out = {'hw': np.array([h, w]), 'img_seed': np.array(img_seed)}
    on = model.out_net
    rec = {}
    dg = on.discrete_gaussian
    orig_bi, orig_c = dg.build_indexes, dg.compress

    def bi(sc):
        idx = orig_bi(sc)
        rec['scale'], rec['indexes'] = npf(sc), npf(idx).astype(np.uint8)
        return idx

    def comp(x, indexes, means=None):
        rec['pm'] = npf(means)
        rec['symbols'] = npf(dg.quantize(x, 'symbols', means)).astype(np.int32)
        s = orig_c(x, indexes, means=means)
        rec['string'] = np.frombuffer(s[0], dtype=np.uint8)
        return s
    dg.build_indexes, dg.compress = bi, comp
    # the raw conv_mean / conv_scale outputs behind pm / scale (_preapre_codec, :69-79): what the guard-band check of a flipped
    # per-pixel mean (round(m * 127.5 + 127.5)) or scale index needs
    hm = on.conv_mean.register_forward_hook(lambda mod, a, o: rec.__setitem__('raw_mean', npf(o)))
    hs = on.conv_scale.register_forward_hook(lambda mod, a, o: rec.__setitem__('raw_logscale', npf(o)))
    obj = model.compress(im)
    hm.remove(); hs.remove()
    dg.build_indexes, dg.compress = orig_bi, orig_c
    for k, v in rec.items():
        out[f'out.{k}'] = v
    for i, s in enumerate(obj[:-2]):
        out[f'string{i}'] = np.frombuffer(s[0], dtype=np.uint8)
    out['smallest'] = np.array(obj[-2])
    out['pickle_bytes'] = np.array(len(pickle.dumps(obj + [(h, w)])))
    xhat = model.decompress(obj)
    out['xhat'] = npf(xhat)
    back = torch.round(xhat * 255.0).to(torch.uint8)[0].permute(1, 2, 0).numpy()
    out['lossless'] = np.array(bool((back == u8).all()))
    out['scale_table'] = npf(dg.scale_table)
    print('qres34m_lossless', tag, 'latent bytes', sum(len(s[0]) for s in obj[:-2]), 'pixel-stream bytes', len(obj[-1][0]),
          'bpp', 8 * (sum(len(s[0]) for s in obj[:-2]) + len(obj[-1][0])) / (h * w), 'lossless', bool(out['lossless']),
          'sym range', int(rec['symbols'].min()), int(rec['symbols'].max()), 'idx range', int(rec['indexes'].min()), int(rec['indexes'].max()))
    np.savez_compressed(os.path.join(OUT, f'qres34m_lossless_{tag}.npz'), **out)


def main_qres():
    model = lvae.get_model('qres34m')
    load_seeded(model, 0)
    model.eval()
    model.compress_mode()
    print('qres34m params', sum(p.numel() for p in model.parameters()) / 1e6, 'entries', len(model.state_dict()))
    dg = model.decoder.dec_blocks[0].discrete_gaussian
    np.savez_compressed(os.path.join(OUT, 'gaussian_conditional_tables.npz'), scale_table=npf(dg.scale_table),
                        quantized_cdf=npf(dg._quantized_cdf), cdf_length=npf(dg._cdf_length), offset=npf(dg._offset))
    golden_qres(model, 64, 64, '64x64')
    golden_qres(model, 128, 192, '128x192', img_seed=1)
    with open(os.path.join(OUT, 'qres34m_state_keys.json'), 'w') as f:
        json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, f)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'qres':
        return main_qres()
    if len(sys.argv) > 1 and sys.argv[1] == 'full':
        # the size BASELINE.json's metric is quoted on (configs 2 and 3): one 512x768 image per model, lambda = 2048 for qarv_base
        # (qarv/model.py:516-557 at full size: symbols, indexes, pm / pv / qm of every block, the bitstream, x_hat)
        model = lvae.get_model('qarv_base')
        load_seeded(model, 0)
        model.eval()
        model.compress_mode()
        golden_qarv(model, 512, 768, [2048.0], '512x768', img_seed=7, full_features=False)
        model = lvae.get_model('qres34m')
        load_seeded(model, 0)
        model.eval()
        model.compress_mode()
        return golden_qres(model, 512, 768, '512x768', img_seed=7)
    if len(sys.argv) > 1 and sys.argv[1] == 'qres17m':
        model = lvae.get_model('qres17m')
        load_seeded(model, 0)
        model.eval()
        model.compress_mode()
        with open(os.path.join(OUT, 'qres17m_state_keys.json'), 'w') as f:
            json.dump({k: list(v.shape) for k, v in model.state_dict().items() if 'discrete_gaussian' not in k}, f)
        return golden_qres(model, 64, 128, '64x128', model_name='qres17m')
    if len(sys.argv) > 1 and sys.argv[1] == 'lossless':
        model = lvae.get_model('qres34m_lossless')
        load_seeded(model, 0)
        model.eval()
        model.compress_mode()
        with open(os.path.join(OUT, 'qres34m_lossless_state_keys.json'), 'w') as f:
            json.dump({k: list(v.shape) for k, v in model.state_dict().items() if 'discrete_gaussian' not in k}, f)
        return golden_qres_lossless(model, 64, 128, '64x128')
    if len(sys.argv) > 1 and sys.argv[1] == 'progressive':
        model = lvae.get_model('qarv_base')
        load_seeded(model, 0)
        model.eval()
        return golden_progressive(model, 64, 128, 16.0, '64x128')
    if len(sys.argv) > 1 and sys.argv[1] == 'robust':
        model = lvae.get_model('qarv_base')
        load_seeded(model, 0)
        model.eval()
        return golden_robust(model, 64, 64, 64.0, '64x64', img_seed=5)
    golden_pack()
    golden_cnx_block()
    model = lvae.get_model('qarv_base')
    load_seeded(model, 0)
    model.eval()
    model.compress_mode()
    n = sum(p.numel() for p in model.parameters())
    print('qarv_base params', n / 1e6, 'entries', len(model.state_dict()))
    golden_tables(model)
    golden_qarv(model, 64, 64, [2048.0, 16.0, 256.0], '64x64')
    golden_qarv(model, 128, 192, [2048.0, 64.0], '128x192', img_seed=1, full_features=False)
    golden_imcoding(model)


if __name__ == '__main__':
    main()
