"""Round 6: the calibrated coder workload (lossy-vae_amd/coder_workloads.py) against the bench's 'typical' synthetic streams.
Prints the streams' statistics (per latent block: mean sigma, mode hit rate, escapes), checks the round trip, and times
decompress_batch on both kinds of strings + single-stream host decode ns/symbol.   usage: python tools/r6_calibrated.py [B H W] [--precision P]"""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import numpy as np, torch
import bench, coder_workloads as cw
from lvae.models.entropy_coding import rans_decode_streams, rans_encode_streams
from lvae.utils import coding

args = [a for a in sys.argv[1:] if not a.startswith('--')]
B, H, W = (int(a) for a in args[:3]) if len(args) >= 3 else (8, 512, 768)
prec = sys.argv[sys.argv.index('--precision') + 1] if '--precision' in sys.argv else 'f16x2'
steps = 15
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
model.set_gemm_precision(prec)
ims = bench.synth_batch(B, H, W, 0).to(dev)
os.system('lscpu | egrep "Model name|^CPU\\(s\\)|Thread|MHz|L1d|L2" 1>&2')

def timed_dec(strings, n=steps):
    for _ in range(3):
        model.decompress_batch(strings); torch.cuda.synchronize(dev)
    t0 = time.time()
    for _ in range(n):
        out = model.decompress_batch(strings); torch.cuda.synchronize(dev)
    return (time.time() - t0) / n * 1e3, out

def timed_enc(x, n=steps):
    for _ in range(3):
        model.compress_batch(x); torch.cuda.synchronize(dev)
    t0 = time.time()
    for _ in range(n):
        s = model.compress_batch(x); torch.cuda.synchronize(dev)
    return (time.time() - t0) / n * 1e3, s

def single_stream_ns(tables, sym, idx):
    enc = rans_encode_streams(tables, [sym], [idx], 1)
    out = np.empty_like(sym)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); rans_decode_streams(tables, enc, [idx], [out], 1); best = min(best, time.perf_counter() - t0)
    assert np.array_equal(out, sym)
    return best / sym.size * 1e9

tables = model._dg().host_tables()
res = {'B': B, 'H': H, 'W': W, 'precision': prec}
# --- typical (the bench's headline workload)
enc_ms, strings = timed_enc(ims)
dec_ms, out = timed_dec(strings)
model.pipeline_groups, g0 = 1, model.pipeline_groups
model.decompress_batch(strings); torch.cuda.synchronize(dev)
pl = next(p for k, p in model._plans.items() if k[0] == 'dec' and k[1] == B and k[4] == 0 and k[-1] == model._prec)
st = cw.stream_stats(tables, pl.sym_np.copy(), pl.idx_np.copy())
li_big = int(np.argmax([z * hw for z, hw in pl.lat_shapes])); z, hw = pl.lat_shapes[li_big]; o = pl.idx_off[li_big]
ns_typ = single_stream_ns(tables, pl.sym_np[o:o + z * hw].copy(), pl.idx_np[o:o + z * hw].copy())
model.pipeline_groups = g0
res['typical'] = dict(enc_ms=round(enc_ms, 3), dec_ms=round(dec_ms, 3), mpx_s=round(B * H * W / (enc_ms + dec_ms) / 1e3, 2), mode_hit_rate=round(st['mode_hit_rate'], 4),
                      escape_rate=round(st['escape_rate'], 5), bits_per_symbol=round(st['ideal_bits'] / st['symbols'], 3), ns_per_symbol_single_stream=round(ns_typ, 2))
print('typical', json.dumps(res['typical']), flush=True)
# --- calibrated
cal, xhat, cst, (syms, idxs) = cw.calibrated_strings(model, B, H // 64, W // 64, seed=1)
dec_ms, out = timed_dec(cal)
assert torch.equal(out, xhat), f'calibrated strings do not decode to the sampled reconstruction: max diff {(out - xhat).abs().max()}'
enc_ms, s2 = timed_enc(xhat)
ns_cal = single_stream_ns(tables, np.ascontiguousarray(syms[li_big][0]), np.ascontiguousarray(idxs[li_big][0]))
res['calibrated'] = dict(enc_ms=round(enc_ms, 3), dec_ms=round(dec_ms, 3), mpx_s=round(B * H * W / (enc_ms + dec_ms) / 1e3, 2), mode_hit_rate=round(cst['mode_hit_rate'], 4),
                         escape_rate=round(cst['escape_rate'], 5), bits_per_symbol=round(cst['bits_per_symbol'], 3), coded_over_ideal=round(cst['coded_over_ideal'], 5),
                         coded_over_entropy=round(cst['coded_over_entropy'], 5), bpp=round(cst['bpp'], 4), ns_per_symbol_single_stream=round(ns_cal, 2), per_block=cst['per_block'])
print('calibrated', json.dumps(res['calibrated']), flush=True)
# index histogram of the calibrated stream (which table rows a stream touches)
hist = np.bincount(np.concatenate([i.reshape(-1) for i in idxs]), minlength=64)
print('index histogram', hist.tolist(), flush=True)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(REPO, 'gpurun_out', f'r6_calibrated_b{B}_{H}x{W}_{prec}.json'), 'w'), indent=1)
# keep one image's streams as a host-side fixture for decoder tuning on the build box
np.savez_compressed(os.path.join(REPO, 'gpurun_out', f'r6_calibrated_streams_b{B}_{H}x{W}.npz'),
                    **{f'sym{li}': syms[li][0] for li in range(len(syms))}, **{f'idx{li}': idxs[li][0] for li in range(len(idxs))})
