#!/usr/bin/env python
"""bench.py -- headline benchmark: Mpixels/s of encode+decode (qarv_base, batch of 8 synthetic 512x768 images per
GPU) with the speedtest-lvae.py protocol (/root/reference/scripts/speedtest-lvae.py:13-44): image tensors already
resident in HBM, `compress` then `decompress`, device sync after each phase, host rANS coding INSIDE the timed region.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the hot path over one batch: compress_batch(8 images) + decompress_batch(8 strings).
Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      -- the dominant kernel family (the PLAIN channel-mixing GEMMs: MLP fc1/fc2 + 1x1 convs, 87% of the path's FLOPs):
                   algorithmic FLOPs of those launches / their HIP-event-measured durations (extra steps after the timed region on the timed
                   region's own plans, the pipeline groups replayed one after the other).  Default arithmetic f16x2: against 2500/3 = 833.3 TFLOP/s (three fp16 MFMAs per fp32-accurate
                   product step); --precision bf16x3: 2500/6 = 416.7; --precision fp32: the 157.3 TFLOP/s fp32 MFMA peak; bf16 / fp8: HBM-bound,
                   algorithmic bytes against 8 TB/s.  `traffic` (HBM bytes per launch) cannot be read from inside this process: it
                   is copied from the committed rocprofv3 --pmc passes of the same command and labelled so (`traffic_source`);
  roofline_e2e  -- the whole step: algorithmic GEMM FLOPs of one step / ms_per_step against both matrix peaks;
  fp32_mfma_mode_value -- the same workload with the exact fp32 MFMA arithmetic (a few extra steps after the timed region);
  config5_value -- BASELINE.json configs[4]: the reduced-precision mode (bf16 storage + MX-fp8 MFMA) on 4 x 1216x1216, Mpixels/s
                   (a few extra steps after the timed region; details in `config5`);
  coder_workloads -- the step on three coder workloads x three operating points (round 6): 'typical' (the headline's streams: 99.5 % mode
                   symbols), 'calibrated' (latents drawn from the model's own discretised prior: lossy-vae_amd/coder_workloads.py -- the
                   statistics a trained model's streams have against its tables) and 'worst_case' (wide-profile weights on uniform-noise
                   images: every table row, escapes), at the headline's batch, for ONE image and on config 5; per row enc / dec ms, Mpixels/s,
                   mode hit rate, escape rate, single-stream decode ns per symbol;
  other_sizes   -- the headline arithmetic at 4 x 1216x1216 and 2 x 1408x2048 (the other image sizes BASELINE.json names);
  roofline.by_kernel -- the dominant family per kernel (launches per step, us per launch, GFLOP and algorithmic MB per launch, TFLOP/s);
  host_coder    -- rANS encode / decode rate of the native host coder on this step's symbols (Msymbols/s, threads);
  cpu_baseline  -- the CPU oracle (oracle/qarv_oracle.py: the reference's op graph on PyTorch CPU + the plain-C restatement of
                   CompressAI's coder, fed with arrays) on the node's PHYSICAL cores: cores/8 oracle processes of 8 torch threads each on
                   disjoint cores, coding different images at once; `cpu_baseline_8core` = one such process alone (comparable with the
                   README's 10700K figure); bounded samples of the same workload (~10-30 s).
Weights are seeded random-init of the qarv_base architecture (no network for checkpoints); data is synthetic.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PROFILE = os.environ.get('LVAE_BENCH_PROFILE', 'typical')   # seeded-weight profile (lossy-vae_amd/seeded_init.py)
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def build_model(device):
    import lvae
    import seeded_init
    m = lvae.get_model('qarv_base')
    sd = m.state_dict()
    for k in list(sd.keys()):
        a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile=PROFILE)
        if a is not None:
            sd[k] = torch.from_numpy(a)
    m.load_state_dict(sd)
    m = m.to(device)
    m.eval()
    m.compress_mode()
    return m, sd


def synth_batch(B, H, W, rank):
    import seeded_init
    ims = [seeded_init.synthetic_image_u8(H, W, seed=1000 + rank * 64 + i) for i in range(B)]
    x = torch.from_numpy(np.stack(ims)).permute(0, 3, 1, 2).float().div(255)
    return x.contiguous()


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are launched on (torch's current stream)."""
    def __init__(self):
        self.pairs = []

    def wrap(self, plan, pred):
        """Return a run(lo,hi) replacement for `plan` that brackets ops selected by pred(fn, args, label) with events."""
        timer = self
        ops = plan.ops

        def run(lo=0, hi=None, stream=None):
            import ctypes
            s = torch.cuda.current_stream(plan.device)
            sp = ctypes.c_void_p(s.cuda_stream)
            for i, (fn, args, label, _side) in enumerate(ops[lo:hi]):
                if not callable(fn):                     # stream-ordering entry of a two-stream plan: everything runs on ONE stream here
                    continue
                sel = pred(fn, args, label)
                if sel:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
                rc = fn(*args, sp)
                if rc != 0:
                    raise RuntimeError(f'{label}: rc={rc}')
                if sel:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
                    timer.pairs.append((e0, e1, label, fn, args))
        return run

    def summary(self):
        tot = 0.0
        for e0, e1, *_ in self.pairs:
            tot += e0.elapsed_time(e1)
        return tot, len(self.pairs)


def launch_work(fn, a):
    """(algorithmic flop, algorithmic HBM bytes) of one recorded GEMM-family launch: a GEMM reads A and writes the result (+ the
    residual rows) once and the weights once; a fused fc1 -> GELU -> fc2 launch (csrc/mlp_h2c.hip) counts as its two GEMMs and moves
    y + residual + result once (4 bytes per element each) and both weight matrices -- the hidden map never leaves the CU."""
    import ctypes
    from lvae import _native as nat
    if fn is nat.lib().lvae_mlp_h2f:
        m = ctypes.cast(a[0], ctypes.POINTER(nat.MlpDesc)).contents
        return 4.0 * m.M * m.C * m.hid, 12.0 * m.M * m.C + 8.0 * m.C * m.hid
    if fn is nat.lib().lvae_mlp_sk:             # fused small-map MLP + its reduce launch: the same two GEMMs (the split-K planes are an artefact)
        m = ctypes.cast(a[0], ctypes.POINTER(nat.MlpSkDesc)).contents
        return 4.0 * m.M * m.C * m.hid, 12.0 * m.M * m.C + 8.0 * m.C * m.hid
    d = ctypes.cast(a[0], ctypes.POINTER(nat.GemmDesc)).contents
    ea = 2 if d.a_bf16 else 4
    eo = 2 if d.out_bf16 else 4
    ew = {0: 4, 1: 2, 2: 6, 3: 1, 4: 4}[d.prec]
    return 2.0 * d.M * d.N * d.K, float(ea * d.M * d.K + ew * d.N * d.K + eo * d.M * d.N * (2 if d.epi in (2, 3) else 1))


def launch_class(fn, a):
    """Which kernel of the family a recorded launch runs (for the per-kernel rows of `roofline.by_kernel`)."""
    import ctypes
    from lvae import _native as nat
    if fn is nat.lib().lvae_mlp_h2f:
        m = ctypes.cast(a[0], ctypes.POINTER(nat.MlpDesc)).contents
        return f'mlp_h2c<{m.C}, {m.hid}> (fused fc1 -> GELU -> fc2)'
    if fn is nat.lib().lvae_mlp_sk:
        return 'mlp_sk + splitk_reduce (fused small-map MLP, split-K contract)'
    d = ctypes.cast(a[0], ctypes.POINTER(nat.GemmDesc)).contents
    if d.prec != 4:
        return f'prec {d.prec} GEMM'
    if d.a_h2:
        return 'gemm_h2p FOLD (pre-split operands, serial split-K)' if d.ksplit > 1 else 'gemm_h2p (pre-split operands)'
    if d.ksplit > 1 and d.defer_reduce:       # prior / posterior heads: the planes are summed by lvae_prior_index_sk_f32 / lvae_quantize_sk_f32
        return 'gemm_h2 split-K, planes summed by the consumer launch (prior / posterior heads)'
    return 'gemm_h2 (fp32 A split in the main loop)' + (' + split-K reduce' if d.ksplit > 1 else '')


def executed_ops(key, pl):
    """The launches of a plan that a step really issues: an encode stops behind the last latent block's quantize launch (the
    reference's CompresionStopFlag, qarv/model.py:310-312), a decode runs the whole plan."""
    return pl.ops[:pl.qcuts[-1]] if key[0].startswith('enc') and pl.qcuts else pl.ops


def roofline_pass(model, dev, plans, n_steps, step_fn, pred):
    """Replay `plans` -- the (key, plan) pairs the timed region ran -- for n_steps extra steps with the pipeline groups one after the
    other (model.serial_groups) and the launches issued one by one from Python, every launch selected by pred(fn, args, label)
    bracketed by HIP events on its own stream.  -> (ms of the selected launches, their number, their algorithmic flop, their
    algorithmic HBM bytes, selected launches per step according to the plans)."""
    timer = KernelTimer()
    expected = sum(1 for k, pl in plans for fn, a, label, _s in executed_ops(k, pl) if callable(fn) and pred(fn, a, label))
    saved = (model.native_group_loops, model.serial_groups)
    # (the per-launch events need the plans replayed launch by launch from Python: the product path runs a group's whole
    #  encode / decode as one native call -- lvae_encode_blocks / lvae_decode_blocks -- which has no hook per launch)
    model.native_group_loops, model.serial_groups = False, True
    n_plans = len(model._plans)
    try:
        for _, pl in plans:
            pl.run = timer.wrap(pl, pred)
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize(dev)
    finally:
        for _, pl in plans:
            if 'run' in pl.__dict__:
                del pl.run                                   # back to Plan.run
        model.native_group_loops, model.serial_groups = saved
    assert len(model._plans) == n_plans, 'the roofline pass must run the plans of the timed region, not build others'
    ms, n = timer.summary()
    flops = bytes_ = 0.0
    by = {}
    for e0, e1, _label, fn, a in timer.pairs:
        f, b = launch_work(fn, a)
        flops += f
        bytes_ += b
        c = by.setdefault(launch_class(fn, a), [0, 0.0, 0.0, 0.0])
        c[0] += 1; c[1] += e0.elapsed_time(e1); c[2] += f; c[3] += b
    roofline_pass.by_kernel = {k: {'launches_per_step': v[0] // max(1, n_steps), 'avg_launch_us': round(v[1] * 1e3 / v[0], 2),
                                   'gflop_per_launch': round(v[2] / v[0] / 1e9, 3), 'alg_mbytes_per_launch': round(v[3] / v[0] / 1e6, 2),
                                   'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else 0.0}
                               for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}
    return ms, n, flops, bytes_, expected


def attach_traffic(roof, precision, B, H, W):
    """HBM bytes per launch of the family: NOT measured in this run (hardware counters cannot be read from inside the process);
    copied from the committed rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE passes of the same command."""
    files = {('f16x2', 8, 512, 768): ['r06_pmc_gemm_traffic.json'], ('bf16x3', 8, 512, 768): ['r02_pmc_gemm_traffic.json'],
             ('fp8', 4, 1216, 1216): ['r06_pmc_gemm_traffic_fp8_1216.json'], ('fp8', 8, 512, 768): ['r02_pmc_gemm_traffic_fp8.json']}
    for f in files.get((precision, B, H, W), []):
        tp = os.path.join(REPO, 'profiles', f)
        if os.path.exists(tp):
            tj = json.load(open(tp))
            roof['traffic'] = round(tj['hbm_mb_per_launch_corrected'] * 2 ** 20)      # (the counters are KiB: the file's "MB" are MiB)
            roof['traffic_source'] = (f'NOT measured in this run: bytes per launch from the committed rocprofv3 --pmc passes of this '
                                      f"command (profiles/{f}: FETCH_SIZE x2 + WRITE_SIZE over {tj['launches']} launches "
                                      'of this kernel family; WRITE_SIZE calibrated in round 6 on known byte counts in the store patterns of this library: ratio 1.000, '
                                      'profiles/r06_write_size_calibration.txt); producer outputs still resident in the 256 MiB Infinity Cache are not '
                                      'counted by the memory-side counters')
            return


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _cpu_worker(args):
    """One worker process of cpu_baseline: its own oracle on `threads` torch threads, pinned to its own cores."""
    idx, threads, H, W, n_images, start_evt, ready_q, done_q, cores = args
    import seeded_init
    from oracle import compressai_semantics, qarv_oracle
    if cores:
        try:
            os.sched_setaffinity(0, cores)
        except OSError:
            pass
    torch.set_num_threads(threads)
    compressai_semantics.EntropyModel.array_io = True
    arch = qarv_oracle.qarv_base_arch()
    sd = seeded_init.seeded_state_dict(qarv_oracle.qarv_param_shapes(arch), seed=0, profile=PROFILE)
    orc = qarv_oracle.QarvOracle(sd)
    orc.compress_mode()
    ims = synth_batch(n_images + 1, H, W, rank=100 + idx)
    s = orc.compress(ims[0:1]); orc.decompress(s)          # warm-up
    ready_q.put(idx)
    start_evt.wait()
    t0 = time.time()
    for i in range(1, n_images + 1):
        s = orc.compress(ims[i:i + 1])
        orc.decompress(s)
    done_q.put((idx, time.time() - t0))


def cpu_baseline(sd, H, W, n_images, threads, procs=1):
    """Bounded sample of the same workload on the host: oracle enc+dec of HxW images (after 1 warm-up image each); the plain-C coder is
    fed with numpy arrays (array_io), not CompressAI's Python lists.  procs == 1: one oracle with torch.set_num_threads(threads) --
    how the reference itself would be run; procs > 1: `procs` independent oracle PROCESSES of `threads` torch threads each on disjoint
    cores, images processed concurrently -- the most a node's cores give this PyTorch op graph (one 128-thread process is slower than
    an 8-thread one: the layers are too small to scale)."""
    if procs > 1:
        import multiprocessing as mp
        ctx = mp.get_context('spawn')
        start_evt, ready_q, done_q = ctx.Event(), ctx.Queue(), ctx.Queue()
        allowed = sorted(os.sched_getaffinity(0))
        per = max(1, len(allowed) // procs)
        ps = [ctx.Process(target=_cpu_worker, args=((i, threads, H, W, n_images, start_evt, ready_q, done_q, allowed[i * per:(i + 1) * per]),))
              for i in range(procs)]
        for p in ps:
            p.start()
        for _ in ps:
            ready_q.get(timeout=600)
        t0 = time.time()
        start_evt.set()
        per_proc = [done_q.get(timeout=600)[1] for _ in ps]
        dt = time.time() - t0
        for p in ps:
            p.join(timeout=60)
        return {'value': round(procs * n_images * H * W / dt / 1e6, 4), 'unit': 'Mpixels/s', 'cores': int(procs * threads), 'kind': 'port',
                'sample': f'{procs} oracle processes x {threads} torch threads on disjoint cores, {n_images} synthetic {H}x{W} images enc+dec '
                          f'each (after 1 warm-up image), oracle/qarv_oracle.py: PyTorch-CPU fp32 op graph + plain-C CompressAI-style rANS fed '
                          f'with arrays; wall {dt:.1f} s (slowest process {max(per_proc):.1f} s)'}
    from oracle import compressai_semantics, qarv_oracle
    compressai_semantics.EntropyModel.array_io = True
    orc = qarv_oracle.QarvOracle({k: v for k, v in sd.items()})
    orc.compress_mode()
    torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    ims = synth_batch(n_images + 1, H, W, rank=99)
    s = orc.compress(ims[0:1]); orc.decompress(s)          # warm-up
    t0 = time.time()
    for i in range(1, n_images + 1):
        s = orc.compress(ims[i:i + 1])
        orc.decompress(s)
    dt = time.time() - t0
    return {'value': round(n_images * H * W / dt / 1e6, 4), 'unit': 'Mpixels/s', 'cores': int(cores), 'kind': 'port',
            'sample': f'{n_images} synthetic {H}x{W} images enc+dec (after 1 warm-up image), oracle/qarv_oracle.py: PyTorch-CPU '
                      f'fp32 op graph + plain-C CompressAI-style rANS fed with arrays, {dt:.1f} s'}


def host_coder_rate(model, strings, B, H, W):
    """rANS rate of the native host coder alone, on the symbols of the benchmark's own batch (decode then re-encode of every
    stream, all coder threads): Msymbols/s each way."""
    import numpy as np
    from lvae.models.entropy_coding import rans_decode_streams, rans_encode_streams
    from lvae.utils import coding
    tables = model._dg().host_tables()
    pl = next(p for k, p in model._plans.items() if k[0] == 'dec' and k[1] == B and k[4] == 0 and k[-1] == model._prec)
    shapes = pl.lat_shapes                                     # (z, hw) per latent block
    streams, idxs, outs = [], [], []
    # indexes are not in the container: decode needs them from the GPU; use the plan's host mirror of the LAST decode (group 0)
    n_img = pl.B
    for b in range(n_img):
        per = coding.unpack_byte_string(strings[b][10:])
        for li, (z, hw) in enumerate(shapes):
            o = pl.idx_off[li] + b * z * hw
            streams.append(per[li]); idxs.append(pl.idx_np[o:o + z * hw].copy()); outs.append(np.empty(z * hw, dtype=np.int32))
    nsym = sum(i.size for i in idxs)
    nthreads = model.coder_threads
    t0 = time.time()
    rans_decode_streams(tables, streams, idxs, outs, nthreads)
    t1 = time.time()
    enc = rans_encode_streams(tables, outs, idxs, nthreads)
    t2 = time.time()
    assert enc == streams
    return {'symbols': int(nsym), 'streams': len(streams), 'threads': int(nthreads or (os.cpu_count() or 0)),
            'decode_msym_s': round(nsym / (t1 - t0) / 1e6, 1), 'encode_msym_s': round(nsym / (t2 - t1) / 1e6, 1),
            'note': 'native C++ coder (lvae_rans_*_batch) alone: one call decoding / re-encoding every stream of one pipeline '
                    "group's images (9 latent blocks each); bytes re-encoded == bytes decoded"}


def _single_stream_ns(tables, sym, idx):
    """ns per symbol of ONE stream on ONE host thread (the latency of the rANS state chain -- what a latent block's decode waits for),
    best of 5; also checks that the stream re-encodes to the bytes that were decoded."""
    from lvae.models.entropy_coding import rans_decode_streams, rans_encode_streams
    enc = rans_encode_streams(tables, [sym], [idx], 1)
    out = np.empty_like(sym)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        rans_decode_streams(tables, enc, [idx], [out], 1)
        best = min(best, time.perf_counter() - t0)
    assert np.array_equal(out, sym) and rans_encode_streams(tables, [out], [idx], 1) == enc, 'bytes re-encoded != bytes decoded'
    return best / max(1, sym.size) * 1e9


def _plan_streams(model, B, prec):
    """(tables, plan, index of the largest latent block) of the decode plan a one-group decode of B images used last."""
    pl = next(p for k, p in model._plans.items() if k[0] == 'dec' and k[1] == B and k[4] == 0 and k[-1] == prec)
    li = int(np.argmax([z * hw for z, hw in pl.lat_shapes]))
    return model._dg().host_tables(), pl, li


def coder_workload_rows(model, dev, ims, precision, steps, kind_rows=('typical', 'calibrated')):
    """One row per coder workload for the batch `ims` under the model's weights: the images as they are ('typical' -- or 'worst_case'
    when the caller passes the wide-profile model and noise images) and 'calibrated' (latents drawn from the model's own discretised
    prior: lossy-vae_amd/coder_workloads.py).  Per row: enc / dec ms per step (the bench's step: compress_batch, sync, decompress_batch,
    sync -- for the calibrated row the encode is that of the sampled reconstruction, the decode that of the calibrated strings), enc+dec
    Mpixels/s, the streams' mode hit rate / escape rate / bits per symbol, single-stream decode ns per symbol of image 0's largest latent
    block (bytes re-encoded == bytes decoded is asserted on it).  enc / dec ms are the median step of the row."""
    import coder_workloads as cw
    B, _, H, W = ims.shape
    groups = model.pipeline_groups

    # (these side rows report the MEDIAN step: with 8-16 steps one scheduling hiccup would otherwise move a row by several per cent;
    #  the headline above is the mean over its timed region, as the contract says)
    def t_enc(x):
        for _ in range(2):
            model.compress_batch(x); torch.cuda.synchronize(dev)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            s = model.compress_batch(x); torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, s

    def t_dec(strings):
        for _ in range(2):
            model.decompress_batch(strings); torch.cuda.synchronize(dev)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            o = model.decompress_batch(strings); torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, o

    rows = {}
    for kind in kind_rows:
        try:
            if kind == 'calibrated':
                strings, xhat, st, (syms, idxs) = cw.calibrated_strings(model, B, H // 64, W // 64, seed=1)
                dec_ms, out = t_dec(strings)
                assert torch.equal(out, xhat), 'calibrated strings do not decode to the sampled reconstruction'
                enc_ms, _ = t_enc(xhat)
                tables = model._dg().host_tables()
                li = int(np.argmax([s.shape[1] for s in syms]))
                ns = _single_stream_ns(tables, np.ascontiguousarray(syms[li][0]), np.ascontiguousarray(idxs[li][0]))
                extra = {'coded_over_table_entropy': round(st['coded_over_entropy'], 5), 'coded_over_ideal': round(st['coded_over_ideal'], 5),
                         'bits_per_symbol': round(st['bits_per_symbol'], 3), 'bpp': round(st['bpp'], 4),
                         'note': "latents drawn from the model's own discretised prior block by block (symbol = round(sigma[index] * N(0,1))), coded by the "
                                 "host coder into the reference's container; decode = decompress_batch of those strings (returns the sampled "
                                 'reconstruction bit for bit), encode = compress_batch of that reconstruction'}
            else:
                enc_ms, strings = t_enc(ims)
                dec_ms, out = t_dec(strings)
                model.pipeline_groups = 1
                model.decompress_batch(strings); torch.cuda.synchronize(dev)
                model.pipeline_groups = groups
                tables, pl, li = _plan_streams(model, B, precision)
                st = cw.stream_stats(tables, pl.sym_np.copy(), pl.idx_np.copy())
                z, hw = pl.lat_shapes[li]; o = pl.idx_off[li]
                ns = _single_stream_ns(tables, pl.sym_np[o:o + z * hw].copy(), pl.idx_np[o:o + z * hw].copy())
                extra = {'bits_per_symbol': round(st['ideal_bits'] / max(1, st['symbols']), 3),
                         'bpp': round(float(np.mean([len(t) * 8 / (H * W) for t in strings])), 4)}
            rows[kind] = {'enc_ms_per_step': round(enc_ms, 3), 'dec_ms_per_step': round(dec_ms, 3), 'value': round(B * H * W / (enc_ms + dec_ms) / 1e3, 3),
                          'unit': 'Mpixels/s', 'steps': steps, 'mode_hit_rate': round(st['mode_hit_rate'], 4), 'escape_rate': round(st['escape_rate'], 5),
                          'symbols_per_image': int(st['symbols'] // B), 'dec_ns_per_symbol_single_stream': round(ns, 2), **extra}
        except Exception as e:                               # a side measurement never loses the headline line
            rows[kind] = {'error': repr(e)}
        finally:
            model.pipeline_groups = groups
    return rows


def build_wide_model(device, coder_threads, precision):
    """The 'wide' seeded-weight profile (posterior x8 / prior x4: symbols span +-20, every table row in use, escapes) -- with uniform-noise
    images the coder's worst case (SURVEY.md 8(d) 'Synthetic inputs')."""
    import lvae
    import seeded_init
    m = lvae.get_model('qarv_base')
    sd = m.state_dict()
    for k in list(sd.keys()):
        a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='wide')
        if a is not None:
            sd[k] = torch.from_numpy(a)
    m.load_state_dict(sd)
    m = m.to(device).eval()
    m.compress_mode()
    m.coder_threads = coder_threads
    m.set_gemm_precision(precision)
    return m


def noise_batch(B, H, W, rank):
    import seeded_init
    ims = [seeded_init.synthetic_image_u8(H, W, seed=2000 + rank * 64 + i, kind='noise') for i in range(B)]
    return torch.from_numpy(np.stack(ims)).permute(0, 3, 1, 2).float().div(255).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)          # (40 since round 6: a host hiccup in one step moves a 20-step mean by several per cent)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--height', type=int, default=512)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--roofline-steps', type=int, default=3)
    ap.add_argument('--serial-groups', action='store_true',
                    help='PROFILING ONLY: the pipeline groups of every call run one after the other (the same plans and launches as the product '
                         'configuration, each launch alone on the GPU) -- what the roofline pass times with HIP events; under rocprofv3 this gives the '
                         'per-kernel durations that `roofline.avg_launch_us` is to be compared with.  The headline value of such a run is NOT the metric.')
    ap.add_argument('--precision', type=str, default='f16x2', choices=['fp32', 'f16x2', 'bf16x3', 'bf16', 'fp8'],
                    help="GEMM arithmetic: f16x2 (default) = fp32-class accuracy from 2-term fp16 splits, 3 fp16 MFMAs per product step; "
                         "bf16x3 = the same from exact 3-term bf16 splits (6 MFMAs); "
                         "fp32 = exact fp32 MFMA; bf16 = operands rounded to bf16; fp8 = BASELINE config 5: bf16 activation storage + "
                         "MX-fp8 MFMA (bf16 / fp8 are not parity paths)")
    ap.add_argument('--cpu-threads', type=int, default=0, help='threads of the main cpu_baseline row (0 = physical cores)')
    ap.add_argument('--fp32-steps', type=int, default=3, help='extra steps in the exact fp32 MFMA mode (0 = skip)')
    ap.add_argument('--b1-steps', type=int, default=20,
                    help="extra single-image steps in the reference's own protocol (scripts/speedtest-lvae.py: one 512x768 image, sync after "
                         'each of compress / decompress) after the timed region -> b1 (0 = skip)')
    ap.add_argument('--qres-steps', type=int, default=5,
                    help='extra steps of BASELINE config 3 (qres34m, 8 x 512x768, seeded weights) after the timed region -> qres34m_value (0 = skip)')
    ap.add_argument('--config5-steps', type=int, default=8,
                    help='extra steps of BASELINE config 5 (fp8 mode, 4 x 1216x1216) after the timed region -> config5_value (0 = skip)')
    ap.add_argument('--size-steps', type=int, default=4,
                    help='extra steps at the other image sizes BASELINE.json names, headline arithmetic -> other_sizes: 4 x 1216x1216 (Tecnick, padded) and '
                         '2 x 1408x2048 (CLIC-sized: config 4 shards such images over the GPUs, this is the per-GPU rate) (0 = skip)')
    ap.add_argument('--coder-steps', type=int, default=10,
                    help='extra steps per coder-workload row after the timed region -> coder_workloads: typical / calibrated (latents drawn from the '
                         "model's own prior) / worst_case (wide-profile weights on uniform-noise images) x (batch of 8, single image, config 5) (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ.get('LVAE_BENCH_SINGLE_GPU_TEST') == '1':     # rehearsal of the N > 1 path on a 1-GPU box: all ranks on cuda:0, gloo
            local_rank = 0
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    B, H, W = args.batch, args.height, args.width
    model, sd = build_model(dev)
    # one rank per GPU shares the node's host cores: every rank's threads (launch threads, rANS coder pool) stay on its slice of
    # the cores of its GPU's NUMA node (skipped when the node topology cannot be read or looks skewed), pool sized to the slice
    ncpu = None
    if world > 1 and os.environ.get('LVAE_BENCH_REHEARSE_HOST') == '1':
        # rehearsal of the N-rank HOST load on a 1-GPU box (tools/host_contention.sh): every rank gets its own 1/N slice of the cores
        # the job may use -- what NUMA pinning gives it on the real node -- although all ranks share cuda:0
        allowed = sorted(os.sched_getaffinity(0))
        per = max(1, len(allowed) // world)
        os.sched_setaffinity(0, allowed[rank * per:(rank + 1) * per])
        ncpu = per
    elif world > 1 and os.environ.get('LVAE_BENCH_SINGLE_GPU_TEST') != '1':
        try:
            from lvae.utils.numa import pin_ranks_collectively
            ncpu = pin_ranks_collectively(local_rank, dist, local_rank, world)
        except Exception as e:                                  # placement is an optimisation: never lose the run over it
            print(f'[rank {rank}] NUMA pinning skipped: {e!r}', file=sys.stderr)
            ncpu = None
    if ncpu:
        model.coder_threads = max(4, ncpu)
    else:
        model.coder_threads = max(8, len(os.sched_getaffinity(0)) // max(1, world))      # the cores this process may use, not the machine's
    model.set_gemm_precision(args.precision)
    model.serial_groups = bool(args.serial_groups)
    ims = synth_batch(B, H, W, rank).to(dev)

    def step():
        strings = model.compress_batch(ims)
        torch.cuda.synchronize(dev)
        t_mid = time.time()
        out = model.decompress_batch(strings)
        torch.cuda.synchronize(dev)
        return strings, out, t_mid

    for _ in range(args.warmup):
        strings, out, _ = step()

    import ctypes as _ct
    from lvae import _native as _nat

    def dominant(fn, a, label):
        """The dominant kernel = every launch of gemm_kernel<*, PLAIN> (the dense channel-mixing GEMMs: MLP fc1/fc2, i.e. 87%
        of the path's FLOPs, plus post_merge / prior / z_proj / upsample 1x1 convs) -- one rocprofv3 kernel-name family."""
        if fn in (_nat.lib().lvae_mlp_h2f, _nat.lib().lvae_mlp_sk):       # fc1 -> GELU -> fc2 of a block as ONE launch (csrc/mlp_h2c.hip; mlp_sk.hip + its reduce): same family, same MFMA stream
            return True
        if fn is not _nat.lib().lvae_gemm_f32:
            return False
        return _ct.cast(a[0], _ct.POINTER(_nat.GemmDesc)).contents.a_mode == _nat.A_PLAIN

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.time()
    t_enc = 0.0
    per_step = []                                          # (encode s, decode s) of every timed step: diagnostics beside the contract's mean
    for _ in range(args.steps):
        ts = time.time()
        strings, out, t_mid = step()
        t_enc += t_mid - ts
        per_step.append((t_mid - ts, time.time() - t_mid))
    barrier()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt, t_enc], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_enc = float(t[0]), float(t[1])

    # rate / distortion stats of this rank's batch, collated over ranks with one all_gather
    bpp = float(np.mean([len(s) * 8 / (H * W) for s in strings]))
    mse = float((out - ims).square().mean())
    stats = torch.tensor([bpp, mse], dtype=torch.float64, device=dev)
    if dist is not None:
        gathered = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(gathered, stats)
        stats = torch.stack(gathered).mean(0)
    bpp, mse = float(stats[0]), float(stats[1])

    # whole-step algorithmic GEMM FLOPs: every GEMM / fused-MLP launch a step issues from the encode + decode plans of the timed configuration
    timed_plans = [(k, pl) for k, pl in model._plans.items() if k[-1] == args.precision]
    step_gflop = sum(launch_work(fn, a)[0] for k, pl in timed_plans for fn, a, _l, _s in executed_ops(k, pl)
                     if callable(fn) and fn in (_nat.lib().lvae_gemm_f32, _nat.lib().lvae_mlp_h2f, _nat.lib().lvae_mlp_sk)) / 1e9
    ms_step = dt / args.steps * 1e3
    e2e_tf = step_gflop / ms_step                          # GFLOP / ms = TFLOP/s
    roofline_e2e = {
        'gemm_gflop_per_step': round(step_gflop, 1), 'ms_per_step': round(ms_step, 3), 'achieved_tflops': round(e2e_tf, 2),
        'frac_of_f16x2_peak_833.3': round(e2e_tf / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4),
        'frac_of_bf16x3_peak_416.7': round(e2e_tf / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4),
        'frac_of_fp32_mfma_peak_157.3': round(e2e_tf / PEAK_FP32_MFMA_TFLOPS, 4),
        'note': 'algorithmic 2*M*N*K of EVERY GEMM launch of one step (the fused MLP launches counted as their two GEMMs) / wall time of the '
                'step (host rANS, depthwise and pointwise kernels included in the time, not in the FLOPs)'}

    roof = None
    if not args.no_kernel_timing:
        # Roofline pass: the timed region above runs the product configuration -- the pipeline groups' plans on their own HIP streams,
        # launched concurrently from their threads, so an event pair around one launch would also time the other group's kernels.
        # The SAME plans (same kernels, same launch mix: nothing is rebuilt and pipeline_groups is not touched) are therefore replayed
        # in `roofline_steps` extra steps ONE GROUP AFTER THE OTHER (model.serial_groups), launch by launch, every launch of the
        # dominant family bracketed by HIP events on the stream it is launched on.
        ms, n_launch, flops, alg_bytes, per_step_expected = roofline_pass(model, dev, timed_plans, args.roofline_steps, step, dominant)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        common = {'traffic': None, 'traffic_source': None, 'by_kernel': getattr(roofline_pass, 'by_kernel', None), 'launches': n_launch, 'launches_per_step': n_launch // max(1, args.roofline_steps),
                  'timed_plans_launches_per_step': per_step_expected, 'avg_launch_us': round(ms * 1e3 / max(1, n_launch), 2),
                  'gflop_per_launch': round(flops / max(1, n_launch) / 1e9, 3),
                  'alg_mbytes_per_launch': round(alg_bytes / max(1, n_launch) / 1e6, 2),
                  'measured_over': f'{args.roofline_steps} extra steps after the timed region on the timed region\'s OWN plans (pipeline groups of '
                                   f'{", ".join(str(k[1]) for k, _ in timed_plans if k[0] == "enc")} images: the groups replayed one after the other, '
                                   'launch by launch), HIP events around every such launch on its stream'}
        if args.precision in ('bf16', 'fp8'):
            kern = {'bf16': 'gemm_bf16_kernel<Cfg<*>, 0> (PLAIN GEMM launches: operands rounded to bf16 on the bf16 MFMA, fp32 maps in HBM)',
                    'fp8': 'gemm_lp_kernel<TN, 0, *, *> (PLAIN GEMM launches of the reduced-precision mode: bf16 maps in HBM, operands '
                           'quantised to MX-fp8 for v_mfma_scale_f32_32x32x64_f8f6f4)'}[args.precision]
            roof = {'bound': 'hbm', 'kernel': kern, 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                    'tflops_equiv': round(ach, 2), **common}
        elif args.precision == 'fp32':
            roof = {'bound': 'mfma', 'kernel': 'gemm_kernel<Cfg<*>, 0> (PLAIN GEMM launches: MLP fc1/fc2 + 1x1 convs; v_mfma_f32_32x32x2_f32)',
                    'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), **common}
        elif args.precision == 'f16x2':
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0           # fp16 MFMA peak = bf16 MFMA peak; 3 MFMAs per fp32-accurate product step
            roof = {'bound': 'mfma', 'kernel': 'gemm_h2p_kernel<WM, TN, NBUF> (MLP fc1 / fc2, both operands pre-split, LDS-DMA main loop) + mlp_h2c_kernel<C, hid, chunk> '
                                               '(fc1 -> GELU -> fc2 of a block as one launch: counted as its two GEMMs, 4 M C hid flop) + '
                                               'gemm_h2_kernel<TN, *, 0> (the other PLAIN GEMM launches); v_mfma_f32_32x32x16_f16 x 3 cross terms',
                    'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'peak_note': '2500 TFLOP/s dense fp16 MFMA / 3 MFMAs per fp32-accurate product step (hi*hi, hi*lo, lo*hi of a 2-term '
                                 f'fp16 split); the same launches against the bf16x3 roof (416.7): {ach / (PEAK_BF16_MFMA_TFLOPS / 6.0):.3f}, '
                                 f'against the fp32 MFMA peak (157.3): {ach / PEAK_FP32_MFMA_TFLOPS:.3f}', **common}
        else:
            peak = PEAK_BF16_MFMA_TFLOPS / 6.0           # algorithmic fp32 FLOPs vs the dense bf16 MFMA peak / 6 MFMAs per product step
            roof = {'bound': 'mfma', 'kernel': 'gemm_x3k16_kernel<TN> / gemm_x3w8_kernel / gemm_x3_kernel<Cfg<*>, 0> (PLAIN GEMM launches; '
                                               'v_mfma_f32_32x32x16_bf16 x 6 cross terms)',
                    'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'peak_note': '2500 TFLOP/s dense bf16 MFMA / 6 MFMAs per fp32-accurate product step; vs the 157.3 TFLOP/s fp32 MFMA '
                                 f'peak this launch family runs at {ach / PEAK_FP32_MFMA_TFLOPS:.3f}', **common}
        attach_traffic(roof, args.precision, B, H, W)

    fp32_mode = None
    if world == 1 and args.fp32_steps > 0 and args.precision in ('f16x2', 'bf16x3'):
        # the same workload with the exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32), product configuration (two groups)
        model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
        model.set_gemm_precision('fp32')
        step(); step()                  # (plans of the mode are built in the first one; a single warm-up step left the 3 timed ones scattered 65 ... 88)
        torch.cuda.synchronize(dev)
        t1 = time.time()
        for _ in range(args.fp32_steps):
            step()
        torch.cuda.synchronize(dev)
        fp32_mode = round(B * H * W * args.fp32_steps / (time.time() - t1) / 1e6, 3)
        model.set_gemm_precision(args.precision)

    # ... and with the EXACT-operand split arithmetic (3-term bf16, 6 MFMAs per product step: round-1/2 default): what the headline's
    # cheaper 2-term fp16 split bought, visible to the driver
    bf16x3_mode = None
    if world == 1 and args.fp32_steps > 0 and args.precision == 'f16x2':
        model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
        model.set_gemm_precision('bf16x3')
        step(); step()
        torch.cuda.synchronize(dev)
        t1 = time.time()
        for _ in range(args.fp32_steps):
            step()
        torch.cuda.synchronize(dev)
        bf16x3_mode = round(B * H * W * args.fp32_steps / (time.time() - t1) / 1e6, 3)
        model.set_gemm_precision(args.precision)

    # The reference's own protocol (scripts/speedtest-lvae.py:28-44; qarv/model.py:521 -- batch 1 only): ONE 512x768 image, synchronise after
    # compress and after decompress, mean over the steps; its README quotes 0.098 s + 0.061 s on an RTX 3080 Ti for this.
    b1 = None
    if world == 1 and args.b1_steps > 0 and (H, W) == (512, 768) and B > 1:
        try:
            model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
            im1 = ims[:1]
            for _ in range(4):                              # the script's warm-up count
                s1 = model.compress(im1); torch.cuda.synchronize(dev); model.decompress(s1); torch.cuda.synchronize(dev)
            te = td = 0.0
            for _ in range(args.b1_steps):
                ta = time.time()
                s1 = model.compress(im1)
                torch.cuda.synchronize(dev)
                tb = time.time()
                model.decompress(s1)
                torch.cuda.synchronize(dev)
                te += tb - ta; td += time.time() - tb
            te, td = te / args.b1_steps * 1e3, td / args.b1_steps * 1e3
            b1 = {'enc_ms': round(te, 3), 'dec_ms': round(td, 3), 'value': round(H * W / (te + td) / 1e3, 3), 'unit': 'Mpixels/s', 'steps': args.b1_steps,
                  'vs_ref_3080ti_latency': round(159.0 / (te + td), 2),
                  'workload': 'qarv_base ONE 512x768 image, compress + decompress with a synchronisation after each (speedtest-lvae.py protocol), '
                              f'{args.precision}; reference README: 98 + 61 ms on an RTX 3080 Ti'}
        except Exception as e:
            b1 = {'error': repr(e)}

    # BASELINE.json configs[2]: qres34m (12 latent blocks, fixed rate), a batch of 8 x 512x768, seeded weights, same step definition
    qres = None
    if world == 1 and args.qres_steps > 0 and (B, H, W) == (8, 512, 768) and args.precision in ('f16x2', 'bf16x3'):
        try:
            import lvae
            import seeded_init
            qm = lvae.get_model('qres34m')
            qsd = qm.state_dict()
            for k in list(qsd.keys()):
                a = seeded_init.seeded_tensor(k, tuple(qsd[k].shape), 0, profile=PROFILE)
                if a is not None and 'discrete_gaussian' not in k:
                    qsd[k] = torch.from_numpy(a)
            qm.load_state_dict(qsd)
            qm.compress_mode()
            qm = qm.to(dev).eval()
            qm.coder_threads = model.coder_threads
            qm.set_gemm_precision(args.precision)

            def stepq():
                o = qm.compress_batch(ims)
                torch.cuda.synchronize(dev)
                x = qm.decompress_batch(o)
                torch.cuda.synchronize(dev)
                return o, x
            for _ in range(2):
                stepq()
            t1 = time.time()
            for _ in range(args.qres_steps):
                oq, xq = stepq()
            dtq = time.time() - t1
            nbytes = sum(sum(len(t[0]) for t in o if isinstance(t, list)) for o in oq)
            qres = {'value': round(B * H * W * args.qres_steps / dtq / 1e6, 3), 'unit': 'Mpixels/s', 'ms_per_step': round(dtq / args.qres_steps * 1e3, 3),
                    'steps': args.qres_steps, 'bpp': round(nbytes * 8 / (B * H * W), 4),
                    'psnr_db': round(float(-10 * np.log10(float((xq - ims).square().mean()))), 3),
                    'workload': f'qres34m batch=8 512x768 synthetic, compress_batch+decompress_batch, {args.precision}, seeded weights (profile {PROFILE})'}
            del qm, oq, xq
        except Exception as e:
            qres = {'error': repr(e)}

    # BASELINE.json configs[4] beside the headline: the reduced-precision mode (bf16 activation storage + MX-fp8 MFMA GEMMs, operands
    # quantised by their producers) on 4 x 1216x1216 (1200x1200 padded) -- a few steps after the timed region, product configuration
    config5 = None
    if world == 1 and args.config5_steps > 0 and args.precision in ('f16x2', 'bf16x3') and (B, H, W) == (8, 512, 768):
        try:
            model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
            model.set_gemm_precision('fp8')
            ims5 = synth_batch(4, 1216, 1216, rank).to(dev)

            def step5():
                s5 = model.compress_batch(ims5)
                torch.cuda.synchronize(dev)
                tm = time.time()
                o5 = model.decompress_batch(s5)
                torch.cuda.synchronize(dev)
                return s5, o5, tm
            for _ in range(3):
                step5()
            t1 = time.time()
            te5 = 0.0
            for _ in range(args.config5_steps):
                ts = time.time()
                s5, o5, tm = step5()
                te5 += tm - ts
            dt5 = time.time() - t1
            # the mode's dominant family against ITS roof (HBM: bf16 maps, one byte per operand element): the plans these steps ran, the
            # groups one after the other, events around every PLAIN GEMM launch -- the same procedure as the headline's `roofline`
            roof5 = None
            if not args.no_kernel_timing:
                plans5 = [(k, pl) for k, pl in model._plans.items() if k[-1] == 'fp8' and k[2:4] in ((1216, 1216), (19, 19))]
                ms5, n5, fl5, by5, exp5 = roofline_pass(model, dev, plans5, 2, step5, dominant)
                gbs5 = by5 / (ms5 * 1e-3) / 1e9 if ms5 > 0 else 0.0
                roof5 = {'bound': 'hbm', 'kernel': 'gemm_q8_kernel<*> (MLP fc1 / fc2: operands quantised by their producers, LDS-DMA main loop) + gemm_lp_kernel<TN, 0, *, *> '
                                                   '(the other PLAIN GEMM launches: bf16 maps in HBM, MX-fp8 operands for v_mfma_scale_f32_32x32x64_f8f6f4)',
                         'achieved': round(gbs5, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs5 / 8000.0, 4),
                         'tflops_equiv': round(fl5 / (ms5 * 1e-3) / 1e12 if ms5 > 0 else 0.0, 2), 'traffic': None, 'launches': n5,
                         'launches_per_step': n5 // 2, 'timed_plans_launches_per_step': exp5, 'avg_launch_us': round(ms5 * 1e3 / max(1, n5), 2),
                         'alg_mbytes_per_launch': round(by5 / max(1, n5) / 1e6, 2),
                         'measured_over': "2 extra steps on this configuration's own plans (groups replayed one after the other, launch by launch), HIP events around every such launch"}
                attach_traffic(roof5, 'fp8', 4, 1216, 1216)
            # the same workload under the fp32-class arithmetic of the headline: encode and decode ratios separately (the decode half is
            # bound by one image's serial rANS -- 2.3 M symbols -- whatever the GPU does: docs/MEASUREMENT_HISTORY.md 5c)
            model.set_gemm_precision(args.precision)
            for _ in range(2):
                step5()
            t1 = time.time()
            tef = 0.0
            nf = max(4, args.config5_steps // 2)
            for _ in range(nf):
                ts = time.time()
                _, _, tm = step5()
                tef += tm - ts
            dtf = time.time() - t1
            model.set_gemm_precision('fp8')
            e5, d5 = te5 / args.config5_steps * 1e3, (dt5 - te5) / args.config5_steps * 1e3
            ef, df = tef / nf * 1e3, (dtf - tef) / nf * 1e3
            config5 = {'value': round(4 * 1216 * 1216 * args.config5_steps / dt5 / 1e6, 3), 'unit': 'Mpixels/s',
                       'ms_per_step': round(dt5 / args.config5_steps * 1e3, 3), 'steps': args.config5_steps,
                       'enc_ms_per_step': round(e5, 3), 'dec_ms_per_step': round(d5, 3),
                       'fp32_class_same_workload': {'precision': args.precision, 'ms_per_step': round(ef + df, 3), 'enc_ms_per_step': round(ef, 3),
                                                    'dec_ms_per_step': round(df, 3), 'value': round(4 * 1216 * 1216 / (ef + df) / 1e3, 3)},
                       'speedup_vs_fp32_class': {'enc': round(ef / e5, 3), 'dec': round(df / d5, 3), 'enc_dec': round((ef + df) / (e5 + d5), 3)},
                       'workload': 'qarv_base batch=4 1216x1216 (1200x1200 padded) synthetic, compress_batch+decompress_batch, '
                                   "set_gemm_precision('fp8'): bf16 activation storage + MX-fp8 (e4m3 + E8M0) MFMA GEMMs; NOT a parity path",
                       'roofline': roof5,
                       'bpp': round(float(np.mean([len(t) * 8 / (1216 * 1216) for t in s5])), 4),
                       'psnr_db': round(float(-10 * np.log10(float((o5 - ims5).square().mean()))), 3)}
            del ims5, o5
        except Exception as e:                           # never lose the headline line over the side measurement
            config5 = {'error': repr(e)}
        model.set_gemm_precision(args.precision)

    # The other image sizes BASELINE.json names, under the headline's arithmetic: Tecnick (1200x1200 padded to 1216) in a
    # batch of 4 and a CLIC-2022-sized image pair -- config 4 shards such images over the node's GPUs with no data-path collective, so
    # its per-GPU rate is this row's.
    other_sizes = None
    if world == 1 and args.size_steps > 0 and args.precision in ('f16x2', 'bf16x3') and (B, H, W) == (8, 512, 768):
        other_sizes = {}
        model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
        model.set_gemm_precision(args.precision)
        for tag, (b_, h_, w_) in (('b4_1216x1216', (4, 1216, 1216)), ('b2_1408x2048', (2, 1408, 2048))):
            try:
                xs = synth_batch(b_, h_, w_, rank).to(dev)
                for _ in range(2):
                    ss = model.compress_batch(xs); torch.cuda.synchronize(dev); model.decompress_batch(ss); torch.cuda.synchronize(dev)
                te = td = 0.0
                for _ in range(args.size_steps):
                    ta = time.time()
                    ss = model.compress_batch(xs); torch.cuda.synchronize(dev)
                    tb = time.time()
                    oo = model.decompress_batch(ss); torch.cuda.synchronize(dev)
                    te += tb - ta; td += time.time() - tb
                te, td = te / args.size_steps * 1e3, td / args.size_steps * 1e3
                other_sizes[tag] = {'value': round(b_ * h_ * w_ / (te + td) / 1e3, 3), 'unit': 'Mpixels/s', 'enc_ms_per_step': round(te, 3),
                                    'dec_ms_per_step': round(td, 3), 'steps': args.size_steps, 'precision': args.precision,
                                    'bpp': round(float(np.mean([len(t) * 8 / (h_ * w_) for t in ss])), 4),
                                    'psnr_db': round(float(-10 * np.log10(float((oo - xs).square().mean()))), 3),
                                    'workload': f'qarv_base batch={b_} {h_}x{w_} synthetic, compress_batch+decompress_batch, {args.precision}'}
                del xs, oo
            except Exception as e:
                other_sizes[tag] = {'error': repr(e)}

    # The coder's workload matters for the decode half: the headline's seeded weights on natural-like images give streams
    # that are 99.4 % mode symbols -- what the host decoder's most-probable-symbol path is fastest on and what no calibrated model writes.
    # Rows beside it, same step definition: 'calibrated' = latents drawn from the model's own discretised prior (coded size == table entropy),
    # 'worst_case' = the wide weight profile on uniform-noise images (every table row, escapes); each at the headline's batch, for ONE image
    # (the reference's protocol) and on config 5.
    coder_rows = None
    if world == 1 and args.coder_steps > 0 and args.precision in ('f16x2', 'bf16x3') and (B, H, W) == (8, 512, 768):
        coder_rows = {}
        try:
            model.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
            model.set_gemm_precision(args.precision)
            cs = args.coder_steps
            coder_rows['b8_512x768'] = coder_workload_rows(model, dev, ims, args.precision, cs)
            coder_rows['b1_512x768'] = coder_workload_rows(model, dev, ims[:1], args.precision, 2 * cs)
            if args.config5_steps > 0:
                model.set_gemm_precision('fp8')
                ims5 = synth_batch(4, 1216, 1216, rank).to(dev)
                coder_rows['config5_b4_1216x1216_fp8'] = coder_workload_rows(model, dev, ims5, 'fp8', max(3, cs // 2))
                model.set_gemm_precision(args.precision)
                del ims5
            wide = build_wide_model(dev, model.coder_threads, args.precision)
            nz = noise_batch(B, H, W, rank).to(dev)
            coder_rows['b8_512x768']['worst_case'] = coder_workload_rows(wide, dev, nz, args.precision, cs, kind_rows=('typical',))['typical']
            coder_rows['b1_512x768']['worst_case'] = coder_workload_rows(wide, dev, nz[:1], args.precision, 2 * cs, kind_rows=('typical',))['typical']
            if args.config5_steps > 0:
                wide.set_gemm_precision('fp8')
                nz5 = noise_batch(4, 1216, 1216, rank).to(dev)
                coder_rows['config5_b4_1216x1216_fp8']['worst_case'] = coder_workload_rows(wide, dev, nz5, 'fp8', max(3, cs // 2), kind_rows=('typical',))['typical']
                del nz5
            del wide, nz
            coder_rows['note'] = ("rows per operating point: 'typical' = the headline's workload (seeded 'typical' weights on natural-like synthetic images: "
                                  "mode symbols almost throughout), 'calibrated' = latents drawn from the model's own discretised prior (the statistics a "
                                  "trained model's streams have against its tables), 'worst_case' = seeded 'wide' weights on uniform-noise images (all table "
                                  'rows, escapes); value = pixels / (enc_ms + dec_ms)')
        except Exception as e:
            coder_rows['error'] = repr(e)
        model.set_gemm_precision(args.precision)

    coder = None
    if rank == 0:
        try:
            model.pipeline_groups = 1
            strings1, _, _ = step()
            coder = host_coder_rate(model, strings1, B, H, W)
        except Exception as e:                           # diagnostic only: never lose the bench line over it
            coder = {'error': repr(e)}

    if model.timing is not None and (rank == 0 or os.environ.get('LVAE_BENCH_REHEARSE_HOST') == '1'):
        print(f'[rank {rank}] host phase timers (s, all steps incl. warm-up):', {k: round(v, 4) for k, v in model.timing.items()},
              f'coder_threads={model.coder_threads}', file=sys.stderr, flush=True)
    if rank == 0:
        px = world * B * H * W * args.steps
        line = {
            'metric': f'Mpixels/s enc+dec (qarv_base, {H}x{W})', 'value': round(px / dt / 1e6, 3), 'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16x3': 'f32 (exact 3-term bf16 split of every fp32 operand, 6 bf16 MFMAs per product step, '
                                               'fp32 accumulate: fp32-class accuracy, parity-tested)',
                      'f16x2': 'f32 (2-term fp16 split of every fp32 operand, hi + lo*2^-11 = 23 of 24 significant bits; 3 fp16 MFMAs per '
                               'product step in two fp32 accumulators: fp32-class accuracy, parity-tested)',
                      'bf16': 'bf16-mfma (f32 activations/accumulate; NOT the parity path)',
                      'fp8': 'mxfp8-mfma (OCP e4m3 + E8M0 block scales, f32 accumulate) with bf16 activation storage: BASELINE config 5, '
                             'NOT the parity path'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('PROFILING RUN (--serial-groups: NOT the metric) ' if args.serial_groups else '') + f'qarv_base batch={B} {H}x{W} synthetic per GPU, compress_batch+decompress_batch, '
                                   f'HIP kernels ({args.precision}) + host rANS, seeded random-init weights (profile {PROFILE})', 'global_batch': world * B,
                       'parallelism': f'dp{world} (images sharded, no data-path collective)',
                       'lambda': model.default_lmb},
            'enc_ms_per_step': round(t_enc / args.steps * 1e3, 3),
            'dec_ms_per_step': round((dt - t_enc) / args.steps * 1e3, 3),
            # the timed steps one by one (this rank): `value` is the contract's mean over all of them; a host hiccup shows here as max >> median
            'step_spread': {'enc_ms': {'median': round(float(np.median([e for e, _ in per_step])) * 1e3, 3), 'min': round(min(e for e, _ in per_step) * 1e3, 3),
                                       'max': round(max(e for e, _ in per_step) * 1e3, 3)},
                            'dec_ms': {'median': round(float(np.median([d_ for _, d_ in per_step])) * 1e3, 3), 'min': round(min(d_ for _, d_ in per_step) * 1e3, 3),
                                       'max': round(max(d_ for _, d_ in per_step) * 1e3, 3)},
                            'slowest_steps': sorted(range(len(per_step)), key=lambda i: -(per_step[i][0] + per_step[i][1]))[:3]},
            'bpp': round(bpp, 4), 'psnr_db': round(-10 * np.log10(mse), 3),
            'ref_3080ti_mpx_s': 2.47,
            'roofline': roof, 'roofline_e2e': roofline_e2e, 'fp32_mfma_mode_value': fp32_mode, 'bf16x3_mode_value': bf16x3_mode,
            'b1': b1, 'qres34m_value': None if not qres else qres.get('value'), 'qres34m': qres,
            'config5_value': None if not config5 else config5.get('value'), 'config5': config5, 'other_sizes': other_sizes, 'coder_workloads': coder_rows, 'host_coder': coder,
        }
        if world == 1 and not args.no_cpu_baseline:
            phys = args.cpu_threads or min(physical_cores(), len(os.sched_getaffinity(0)))
            # the node's cores as 8-thread oracle processes working on different images at once (a single process scales badly past
            # ~8 threads on these layer sizes: 0.2 Mpixels/s on 128 threads against 0.47 on 8); the single 8-thread figure -- what the
            # reference's README quotes a desktop CPU at -- is kept beside it
            line['cpu_baseline'] = cpu_baseline(sd, H, W, n_images=3, threads=8, procs=max(1, phys // 8))
            line['cpu_baseline_8core'] = cpu_baseline(sd, H, W, n_images=3, threads=8)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
