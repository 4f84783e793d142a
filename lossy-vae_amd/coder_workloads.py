"""Coder workloads for bench.py / tests (measurement infrastructure, like seeded_init.py -- not on the product path).

The reference times its codec on Kodak with TRAINED weights (/root/reference/scripts/speedtest-lvae.py:13-44): the latent
streams the host rANS coder sees there obey the model's own tables (a trained model's posterior statistics are what its prior
predicts, /root/reference/lvae/models/qarv/model.py:94-97,112-113).  There is no network for checkpoints here, and the seeded
random-init weights give a DEGENERATE stream on natural-like images: the posterior means sit on the prior means, so 99.4 % of the
symbols are their row's mode although the tables price the mode at 1.58 bits -- a stream no calibrated
model produces, and one that flatters a decoder with a most-probable-symbol fast path.  Two further workloads therefore:

`calibrated_strings`  -- latents DRAWN FROM THE MODEL'S OWN DISCRETISED PRIOR, block by block down the decoder: after a block's
    prior segment has produced its scale indexes, every symbol is round(sigma[index] * N(0, 1)) -- exactly the distribution
    the row's CDF was built from (lvae_build_gaussian_tables: pmf(k) = Phi((k + .5) / sigma) - Phi((k - .5) / sigma), the tails
    beyond the table's support are the escape symbol) -- the block's symbols are written where the decode plan's dequantize launch
    reads them and the decoder goes on to the next block.  The symbols are then coded by the product's host coder into the reference's
    container.  By construction the coded size equals the table entropy (checked: `stats['coded_over_ideal']`), the mode hit rate is
    what the tables say it is, and decompress_batch() on these strings is a decode of a calibrated stream.
`stream_stats`        -- mode hit rate / escape rate / ideal bits of any set of (symbols, indexes) against the tables.
The 'worst case' of SURVEY.md 8(d) (the 'wide' weight profile on uniform-noise images: 4-35 % escapes) needs no generator: bench.py
builds a second model with seeded_init's 'wide' profile and codes seeded_init.synthetic_image_u8(kind='noise') images.
"""
import struct

import numpy as np
import torch


def _row_model(tables):
    """Per table row: (value of the mode, -log2 P of every own symbol, -log2 P(escape), expected bits per symbol)."""
    qcdf, cdf_len, offset = tables
    n = len(cdf_len)
    out = []
    for r in range(n):
        L = int(cdf_len[r])
        cdf = qcdf[r, :L].astype(np.int64)
        f = np.diff(cdf).astype(np.float64)              # L - 1 entries: own symbols 0 .. L - 3, then the escape symbol
        p = f / 65536.0
        bits = -np.log2(np.maximum(p, 1e-300))
        own = f[:-1]
        mode = int(np.argmax(own))
        # expected code length of the row ~ entropy of its quantised pmf (the escape's bypass nibbles add < 1e-6 bits on these tables)
        ent = float(np.sum(p * bits))
        out.append((mode + int(offset[r]), bits[:-1], float(bits[-1]), ent, int(offset[r]), L - 2))
    return out


def stream_stats(tables, sym, idx):
    """sym int32 / idx uint8 arrays of equal length.  -> dict(symbols, mode_hit_rate, escape_rate, ideal_bits, entropy_bits)."""
    rows = _row_model(tables)
    sym = np.asarray(sym).reshape(-1).astype(np.int64)
    idx = np.asarray(idx).reshape(-1).astype(np.int64)
    hits = esc = 0
    ideal = expect = 0.0
    for r in np.unique(idx):
        mode_val, bits, esc_bits, ent, off, mv = rows[int(r)]
        s = sym[idx == r]
        v = s - off
        inr = (v >= 0) & (v < mv)
        hits += int(np.count_nonzero(s == mode_val))
        esc += int(np.count_nonzero(~inr))
        ideal += float(bits[v[inr]].sum())
        if (~inr).any():
            # escape: the escape symbol + 4-bit nibbles (count nibbles + value nibbles), rans_host.cpp::lvae_rans_encode_with_indexes
            ve = v[~inr]
            raw = np.where(ve < 0, -2 * ve - 1, 2 * (ve - mv)).astype(np.int64)
            nb = np.zeros_like(raw)
            for j in range(8):
                nb += (raw >> (4 * j)) != 0
            ideal += float((esc_bits + 4.0 * (nb + nb // 15 + 1)).sum())
        expect += ent * s.size
    n = int(sym.size)
    return {'symbols': n, 'mode_hit_rate': hits / max(1, n), 'escape_rate': esc / max(1, n), 'ideal_bits': ideal, 'entropy_bits': expect}


@torch.no_grad()
def calibrated_strings(model, B, nH, nW, lmb=None, seed=0):
    """-> (strings, x_hat, stats): `strings` = B byte strings in the reference's container (qarv/model.py:525-528) whose nine latent
    streams hold symbols drawn from the model's own discretised prior (see the module docstring); `x_hat` (B, 3, 64 nH, 64 nW) =
    the reconstruction those latents decode to -- decompress_batch(strings) must return exactly it; `stats` = stream_stats of all
    symbols + coded bytes: 'coded_over_ideal' is (8 * payload bytes) / (sum of -log2 P of the symbols under the tables)."""
    from lvae.models.entropy_coding import rans_encode_streams
    from lvae.utils import coding
    lmb = float(lmb or model.default_lmb)
    dev = model._dummy.device
    with torch.cuda.device(dev):
        model._prepare()
        model._set_lmb(lmb)
        pl = model._plan('dec', B, nH, nW)
        tables = model._dg().host_tables()
        sigma = model._dg().scale_table.detach().cpu().numpy().astype(np.float64)
        rng = np.random.Generator(np.random.Philox(key=int(seed) + 0x5EED))
        lo = 0
        syms, idxs = [], []
        for li, cut in enumerate(pl.cuts):
            pl.run(lo, cut)
            torch.cuda.synchronize(dev)
            z, hw = pl.lat_shapes[li]
            o, cnt = pl.idx_off[li], B * z * hw
            idx = pl.idx_all[o:o + cnt].cpu().numpy()
            sym = np.rint(sigma[idx] * rng.standard_normal(cnt)).astype(np.int32)
            pl.sym_all[o:o + cnt].copy_(torch.from_numpy(sym).to(dev))
            syms.append(sym.reshape(B, z * hw)); idxs.append(idx.reshape(B, z * hw).copy())
            lo = cut
        pl.run(lo, None)
        pl.fetch_status()
        torch.cuda.synchronize(dev)
        # (random latents of an untrained model may run a deep block's activations out of the arithmetic's range: not an error of the
        #  workload generator -- the caller sees it as a failed decode of the strings)
        pl.status.zero_(); pl.status_host.zero_()
        x_hat = pl.out.clone()
    nb = len(pl.cuts)
    sv = [np.ascontiguousarray(syms[li][b]) for li in range(nb) for b in range(B)]
    iv = [np.ascontiguousarray(idxs[li][b]) for li in range(nb) for b in range(B)]
    enc = rans_encode_streams(tables, sv, iv, 0)
    header = struct.pack('f', lmb) + struct.pack('3H', 1, nH, nW)
    strings = [header + coding.pack_byte_strings([enc[li * B + b] for li in range(nb)]) for b in range(B)]
    st = stream_stats(tables, np.concatenate([s.reshape(-1) for s in syms]), np.concatenate([i.reshape(-1) for i in idxs]))
    payload = sum(len(e) for e in enc)
    st['payload_bytes'] = payload
    st['coded_over_ideal'] = 8.0 * payload / max(1.0, st['ideal_bits'])
    st['coded_over_entropy'] = 8.0 * payload / max(1.0, st['entropy_bits'])
    st['bits_per_symbol'] = 8.0 * payload / max(1, st['symbols'])
    st['bpp'] = 8.0 * sum(len(s) for s in strings) / (B * nH * nW * 4096)
    st['per_block'] = [dict(block=li, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in
                            stream_stats(tables, syms[li], idxs[li]).items() if k in ('symbols', 'mode_hit_rate', 'escape_rate')},
                            mean_sigma=round(float(sigma[idxs[li].reshape(-1)].mean()), 3)) for li in range(nb)]
    return strings, x_hat, st, (syms, idxs)
